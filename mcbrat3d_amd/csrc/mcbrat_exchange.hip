// mcbrat_exchange.hip -- the tracing kernel in its photon-exchange form (included by mcbrat_api.hip after
// mcbrat_kernels.hip, whose helpers it uses).
//
// Same physics, same Philox slots and the same arithmetic as trace_kernel -- results are bitwise identical -- but a
// different division of labour.  In trace_kernel a photon stays in one lane for life, and the lane idles whenever
// the rest of its wave is in the other half of the loop: ~64 % of the lanes work in a walk iteration, ~39 % in an
// event phase (128x128x64).  Here a photon lives in an LDS slot of its workgroup (80 bytes) and moves between
// waves at leg boundaries:
//   * a lane that WALKS holds only the leg's geometry in registers (origin, direction, face distances, optical
//     depth).  When its leg ends it writes the end point back to the slot, queues the slot for event processing
//     and takes the next waiting leg from the walk queue -- so walk iterations run with nearly all lanes busy;
//   * whenever 64 collisions / surface hits have queued up, a wave takes them and processes them in one dense pass
//     (optics, absorption tally, roulette, scattering angle, new direction, next free path) and queues the 64 new
//     legs for walking; 64 dead photons are relaunched the same way.
// The queues are bounded multi-producer / multi-consumer rings of slot numbers in LDS (one sequence number per
// cell, as in D. Vyukov's bounded queue): a wave reserves a block of cells with one LDS atomic and each lane
// then fills or drains its own cell.  A slot number is always in exactly one place (a queue, a walking lane, a
// wave's event pass) and there are fewer slots than cells, so a ring never overflows.
// Covers what the flux benches need: dense optical grids (global memory or LDS), the directional source,
// no radiance, no instrumentation; everything else stays with trace_kernel.
#include "mcbrat_device.h"

// Development aid: -DMCBRAT_XSTATS makes every wave report cycles per section (refill, walk, flush, event pass, launch
// pass, idle) and how many lanes each kind of pass served, through DevParams::counters (slots 16..).
#ifdef MCBRAT_XSTATS
#define XSTAMP(i) do { const unsigned long long t_ = clock64(); xs[i] += t_ - xt; xt = t_; } while (0)
#define XCOUNT(i, v) do { xs[i] += (unsigned long long)(v); } while (0)
#else
#define XSTAMP(i) do { } while (0)
#define XCOUNT(i, v) do { } while (0)
#endif

namespace mcbrat {

constexpr int kXCells = 1024;  // cells per ring (a power of two, more than the slots of a workgroup)
enum : unsigned { XK_COLLIDE = 1u, XK_SURFACE = 2u };
// queue control words in LDS: head/tail of the walk, event and dead rings, then the slots still in use
enum : int { XQ_WALK = 0, XQ_EVENT = 2, XQ_DEAD = 4, XQ_LIVE = 6 };

__device__ __forceinline__ unsigned x_load(const unsigned *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void x_store(unsigned *p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// entries in a ring / a control word, as one value for the whole wave (branches on it must be wave-uniform)
__device__ __forceinline__ int xq_count(const unsigned *ctl) {
  return __builtin_amdgcn_readfirstlane((int)(x_load(&ctl[1]) - x_load(&ctl[0])));
}
__device__ __forceinline__ unsigned x_load_uniform(const unsigned *p) {
  return (unsigned)__builtin_amdgcn_readfirstlane((int)x_load(p));
}

// Every active lane appends one slot number.  Whatever the lanes wrote to their slots before is visible to the
// wave that takes them out (release here, acquire in xq_pop).
__device__ __forceinline__ void xq_push(unsigned *ctl, unsigned *cells, bool active, unsigned value, int lane,
                                        unsigned long long laneBelow) {
  const unsigned long long m = __ballot(active);
  if (m == 0ull) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  const int leader = __ffsll((long long)m) - 1;
  unsigned base = 0;
  if (lane == leader) base = atomicAdd(&ctl[1], (unsigned)__popcll(m));
  base = (unsigned)__shfl((int)base, leader);
  if (active) {
    const unsigned pos = base + (unsigned)__popcll(m & laneBelow);
    unsigned *cell = cells + (pos & (kXCells - 1));
    while ((x_load(cell) >> 16) != (pos & 0xffffu)) __builtin_amdgcn_s_sleep(1);  // (free: fewer slots than cells)
    x_store(cell, (((pos + 1u) & 0xffffu) << 16) | value);
  }
}

// Slot numbers for the lanes that ask, lowest lanes first; -1 for a lane that got none.  Takes nothing unless at
// least minTake entries are there.
__device__ __forceinline__ int xq_pop(unsigned *ctl, unsigned *cells, bool want, int minTake, int lane,
                                      unsigned long long laneBelow) {
  const unsigned long long m = __ballot(want);
  if (m == 0ull) return -1;
  const int leader = __ffsll((long long)m) - 1;
  unsigned base = 0, n = 0;
  if (lane == leader) {
    const unsigned nWant = (unsigned)__popcll(m);
    const unsigned need = (unsigned)minTake < nWant ? (unsigned)minTake : nWant;
    unsigned h = x_load(&ctl[0]);
    for (;;) {
      const int avail = (int)(x_load(&ctl[1]) - h);
      n = avail < (int)nWant ? (unsigned)(avail > 0 ? avail : 0) : nWant;
      if (n == 0 || n < need) { n = 0; break; }
      const unsigned old = atomicCAS(&ctl[0], h, h + n);
      if (old == h) { base = h; break; }
      h = old;
    }
  }
  base = (unsigned)__shfl((int)base, leader);
  n = (unsigned)__shfl((int)n, leader);
  int v = -1;
  const unsigned rank = (unsigned)__popcll(m & laneBelow);
  if (want && rank < n) {
    const unsigned pos = base + rank;
    unsigned *cell = cells + (pos & (kXCells - 1));
    unsigned cv;
    while (((cv = x_load(cell)) >> 16) != ((pos + 1u) & 0xffffu)) __builtin_amdgcn_s_sleep(1);  // its producer is writing it
    v = (int)(cv & 0xffffu);
    x_store(cell, ((pos + (unsigned)kXCells) & 0xffffu) << 16);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return v;
}

// Slot record, 5 x 16 bytes:
//   q0 = px, py (double)            q1 = pz (double), dx, dy
//   q2 = dz, tau, ix | iy << 10 | iz << 20 | kind << 30, extinction of the cell
//   q3 = weight, photon id (lo, hi), event number        q4 = batch, uX, uY, uZ (the leg's Philox block)
// A walking lane reads q0-q2 and writes back the end point (q0, pz) and the cell / kind / extinction (q2.zw).
template <int BLOCK, bool TBL_LDS, int PRIV>
__global__ void __launch_bounds__(BLOCK, MCBRAT_MIN_WAVES_PER_SIMD) trace_kernel_x(const DevParams p, const int nSlots,
                                                                                  const int flushLanes) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  // LDS map: as trace_kernel, then [queue control (8 words)] [3 rings of kXCells cells] [slot pool]
  double *s_edge = reinterpret_cast<double *>(smem_raw);
  const int nEdges = p.nx + p.ny + p.nz + 3;
  const int ncol = p.nx * p.ny;
  const int slabLen = PRIV ? (int)p.slabStride : 0;
  long long *s_slab = reinterpret_cast<long long *>(s_edge + nEdges);
  unsigned *s_cursor = reinterpret_cast<unsigned *>(s_slab + slabLen);
  float *s_bgExt = reinterpret_cast<float *>(s_cursor + (PRIV ? 4 : 0));
  constexpr bool gridLds = PRIV == 2;
  const int nvoxS = gridLds ? ncol * p.nz : 0;
  int *s_run = reinterpret_cast<int *>(s_bgExt + ((p.nz + 3) & ~3));
  double *s_runT = reinterpret_cast<double *>(s_run + ((p.nz + 3) & ~3));
  float *s_ext = reinterpret_cast<float *>(s_runT + ((p.nz + 2) & ~1));
  float *s_ssa = s_ext + nvoxS;
  float *s_cum = s_ssa + (size_t)p.nc * nvoxS;
  uint16_t *s_pfi = reinterpret_cast<uint16_t *>(s_cum + (size_t)p.nc * nvoxS);
  float *s_tbl = reinterpret_cast<float *>(s_pfi + (((size_t)p.nc * nvoxS + 1) & ~(size_t)1));
  const size_t exOff = ((size_t)(reinterpret_cast<unsigned char *>(s_tbl + (TBL_LDS ? p.tblTotalFloats : 0)) - smem_raw) + 15) & ~(size_t)15;
  unsigned *s_qctl = reinterpret_cast<unsigned *>(smem_raw + exOff);
  unsigned *s_cells = s_qctl + 8;
  uint4 *s_pool = reinterpret_cast<uint4 *>(s_cells + 3 * kXCells);
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
  if (PRIV) {
    for (int i = threadIdx.x; i < slabLen; i += BLOCK) s_slab[i] = 0;
    if (threadIdx.x == 0) s_cursor[0] = 0;
  }
  for (int i = threadIdx.x; i < p.nz; i += BLOCK) { s_bgExt[i] = p.bgExt[i]; s_run[i] = p.layerRun[i]; }
  for (int i = threadIdx.x; i <= p.nz; i += BLOCK) s_runT[i] = p.layerRunT[i];
  if (gridLds) {
    for (int i = threadIdx.x; i < nvoxS; i += BLOCK) s_ext[i] = p.ext[i];
    for (int i = threadIdx.x; i < p.nc * nvoxS; i += BLOCK) { s_ssa[i] = p.ssa[i]; s_cum[i] = p.cum[i]; s_pfi[i] = p.pfi[i]; }
  }
  const float *__restrict__ tbl = TBL_LDS ? s_tbl : p.tables;
  const int offY = p.nx + 1, offZ = p.nx + p.ny + 2;
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned long long laneBelow = (1ull << lane) - 1ull;
  unsigned *qWalk = s_cells, *qEvent = s_cells + kXCells, *qDead = s_cells + 2 * kXCells;

  unsigned long long unitFirst = 0;
  unsigned unitCount = 0;
  long long *unitSlab = nullptr;
  uint32_t unitBatch = 0;

  // walking lane ---------------------------------------------------------------------------------------
  int state = ST_DEAD;  // ST_DEAD: the lane holds no photon
  int slot = 0;
  double px = 0, py = 0, pz = 0;
  float dx = 0, dy = 0, dz = 1, ivx = 0, ivy = 0, ivz = 0;
  float tnx = 0, tny = 0, tnz = 0, tcur = 0, acc = 0, tau = 0, extCur = 0;
  int ex = 0, ey = 0, ez = 0, cell = 0;
  constexpr bool LAZY = PRIV != 2;
  bool lazy = false;

  auto tryJump = [&]() {  // (trace_kernel: same function)
    const bool up = dz >= 0.0f;
    const int k = ez - offZ - (up ? 1 : 0);
    const int run = s_run[k];
    const int fEnd = up ? (run >> 16) : (run & 0xffff);
    const double dT = up ? s_runT[fEnd] - s_runT[k + 1] : s_runT[k] - s_runT[fEnd];
    const float accEnd = acc + (tnz - tcur) * extCur + (float)dT * fabsf(ivz);
    lazy = true;
    tnx = FLT_MAX; tny = FLT_MAX;
    state = ST_WALK;
    if (accEnd <= tau) {
      acc = accEnd;
      tcur = (float)(s_edge[offZ + fEnd] - pz) * ivz;
      if (up) { ez = offZ + fEnd + 1; state = fEnd == p.nz ? ST_TOP : ST_ENTER; }
      else { ez = offZ + fEnd - 1; state = fEnd == 0 ? ST_SURFACE : ST_ENTER; }
      if (state == ST_SURFACE) extCur = s_bgExt[0];
      tnz = (float)(s_edge[min(max(ez, offZ), offZ + p.nz)] - pz) * ivz;
    }
  };
  auto resolveXY = [&]() {  // (trace_kernel: same function)
    const float tc = state == ST_COLLIDE ? tcur + div_fast(tau - acc, extCur) : tcur;
    const double xw = px + (double)tc * (double)dx, yw = py + (double)tc * (double)dy;
    const int jx = locate_periodic(s_edge, p.nx, p.x0, p.Lx, p.invLx, p.invCellX, p.xyNearUniform != 0, px, xw);
    const int jy = locate_periodic(s_edge + offY, p.ny, p.y0, p.Ly, p.invLy, p.invCellY, p.xyNearUniform != 0, py, yw);
    ex = jx + (dx >= 0.0f ? 1 : 0);
    ey = offY + jy + (dy >= 0.0f ? 1 : 0);
    tnx = ivx != 0.0f ? (float)(s_edge[ex] - px) * ivx : FLT_MAX;
    tny = ivy != 0.0f ? (float)(s_edge[ey] - py) * ivy : FLT_MAX;
    cell = jx + p.nx * (jy + p.ny * (ez - offZ - (dz >= 0.0f ? 1 : 0)));
    lazy = false;
    if (state == ST_ENTER) { extCur = p.ext[cell]; state = ST_WALK; }
  };

#ifdef MCBRAT_XSTATS
  unsigned long long xs[16] = {0}, xt = clock64();
#endif
  for (unsigned long long unit = blockIdx.x;; unit += gridDim.x) {
    if (PRIV) {  // workgroup-uniform: this workgroup's unit = photons [unitFirst, unitFirst + unitCount) of ONE batch
      if (unit >= p.nUnits) break;
      const unsigned long long b = unit / p.unitsPerBatch, s = unit % p.unitsPerBatch;
      const unsigned long long bp = (p.total - b * p.ppb) < p.ppb ? (p.total - b * p.ppb) : p.ppb;
      const unsigned long long lo = (bp * s) / p.unitsPerBatch, hi = (bp * (s + 1)) / p.unitsPerBatch;
      unitFirst = b * p.ppb + lo;
      unitCount = (unsigned)(hi - lo);
      unitSlab = p.slabs + b * p.slabStride;
      unitBatch = (uint32_t)b;
    }
    // all slots start dead: the dead ring holds 0 .. nSlots-1
    for (int i = threadIdx.x; i < 3 * kXCells; i += BLOCK) {
      const unsigned j = (unsigned)(i & (kXCells - 1));
      s_cells[i] = (i >= 2 * kXCells && (int)j < nSlots) ? ((((j + 1u) & 0xffffu) << 16) | j) : (j << 16);
    }
    if (threadIdx.x < 8) s_qctl[threadIdx.x] = (threadIdx.x == XQ_DEAD + 1 || threadIdx.x == XQ_LIVE) ? (unsigned)nSlots : 0u;
    __syncthreads();

    for (;;) {
      bool worked = false;
      XSTAMP(5);
      // ================= walking lanes: take waiting legs =================================================
      {
        const int nIdle = __popcll(__ballot(state == ST_DEAD));
        if (nIdle >= flushLanes || nIdle == kWave) {
          if (xq_count(s_qctl + XQ_WALK) > 0) {
            const int got = xq_pop(s_qctl + XQ_WALK, qWalk, state == ST_DEAD, 1, lane, laneBelow);
            if (got >= 0) {
              slot = got;
              const uint4 q0 = s_pool[5 * slot], q1 = s_pool[5 * slot + 1], q2 = s_pool[5 * slot + 2];
              px = __hiloint2double((int)q0.y, (int)q0.x);
              py = __hiloint2double((int)q0.w, (int)q0.z);
              pz = __hiloint2double((int)q1.y, (int)q1.x);
              dx = __uint_as_float(q1.z); dy = __uint_as_float(q1.w); dz = __uint_as_float(q2.x);
              tau = __uint_as_float(q2.y);
              extCur = __uint_as_float(q2.w);
              const int ix = (int)(q2.z & 1023u), iy = (int)((q2.z >> 10) & 1023u), iz = (int)((q2.z >> 20) & 1023u);
              acc = 0.0f; tcur = 0.0f;
              cell = ix + p.nx * (iy + p.ny * iz);
              // opticalProperties.f95:1690-1712 (trace_kernel: start of a leg)
              ex = ix + (dx >= 0.0f ? 1 : 0);
              ey = offY + iy + (dy >= 0.0f ? 1 : 0);
              ez = offZ + iz + (dz >= 0.0f ? 1 : 0);
              if (fabsf(dz) >= 2.0f * FLT_MIN) { ivz = rcp_fast(dz); tnz = (float)(s_edge[ez] - pz) * ivz; }
              else { ivz = 0.0f; tnz = FLT_MAX; }
              ivx = fabsf(dx) >= 2.0f * FLT_MIN ? rcp_fast(dx) : 0.0f;
              ivy = fabsf(dy) >= 2.0f * FLT_MIN ? rcp_fast(dy) : 0.0f;
              state = ST_WALK;
              lazy = false;
              if (LAZY && p.layerSkip && ivz != 0.0f && s_bgExt[iz] >= 0.0f) {
                tryJump();
                if (state == ST_ENTER) resolveXY();
              } else {
                tnx = ivx != 0.0f ? (float)(s_edge[ex] - px) * ivx : FLT_MAX;
                tny = ivy != 0.0f ? (float)(s_edge[ey] - py) * ivy : FLT_MAX;
              }
            }
            worked = true;
          }
        }
      }
      XSTAMP(0);
      // ================= walk: one voxel face per iteration (trace_kernel's loop) =========================
      {
        const int nStart = __popcll(__ballot(state == ST_WALK));
        if (__ballot(state != ST_DEAD) != 0ull) {  // (a leg can end where it starts: a run of layers taken in one step)
          worked = true;
          const int stopBelow = max(nStart - flushLanes + 1, 1);  // until flushLanes lanes have stopped, or all
          int nWalk = nStart;
          while (nWalk >= stopBelow) {
            XCOUNT(6, 1); XCOUNT(7, nWalk);
            if (state == ST_WALK) {
              const bool yLtX = tny < tnx;
              const float m2 = yLtX ? tny : tnx;
              const bool isZ = tnz < m2;
              const float tmin = isZ ? tnz : m2;
              const float accNew = acc + (tmin - tcur) * extCur;  // :1743
              if (accNew > tau) {
                state = ST_COLLIDE;
              } else {
                acc = accNew;
                tcur = tmin;
                if (isZ) {
                  ez += dz >= 0.0f ? 1 : -1;
                  if (ez > offZ + p.nz) state = ST_TOP;
                  else if (ez < offZ) state = ST_SURFACE;
                  else cell += dz >= 0.0f ? ncol : -ncol;
                } else if (yLtX) {
                  ey += dy >= 0.0f ? 1 : -1;
                  cell += dy >= 0.0f ? p.nx : -p.nx;
                  if (ey > offY + p.ny) { ey = offY + 1; cell -= ncol; py -= p.Ly; }
                  else if (ey < offY) { ey = offY + p.ny - 1; cell += ncol; py += p.Ly; }
                } else {
                  ex += dx >= 0.0f ? 1 : -1;
                  cell += dx >= 0.0f ? 1 : -1;
                  if (ex > p.nx) { ex = 1; cell -= p.nx; px -= p.Lx; }
                  else if (ex < 0) { ex = p.nx - 1; cell += p.nx; px += p.Lx; }
                }
                if (gridLds) extCur = s_ext[cell];
                else if (state == ST_WALK) {
                  const int k = ez - offZ - (dz >= 0.0f ? 1 : 0);
                  const float lv = s_bgExt[k];
                  if (LAZY && p.layerSkip && isZ && (lv >= 0.0f) != lazy) state = lazy ? ST_ENTER : ST_JUMP;
                  extCur = lv;
                  if (lv < 0.0f && state == ST_WALK) extCur = p.ext[cell];
                }
                const bool spaced = isZ ? p.zRegularWalk != 0 : p.xyRegularWalk != 0;
                float tNew = tmin + (isZ ? p.dZf * fabsf(ivz) : (yLtX ? p.dYf * fabsf(ivy) : p.dXf * fabsf(ivx)));
                if (!spaced) {
                  const int eSel = isZ ? min(max(ez, offZ), offZ + p.nz) : (yLtX ? ey : ex);
                  const double edge = s_edge[eSel];
                  const double origin = isZ ? pz : (yLtX ? py : px);
                  const float iv = isZ ? ivz : (yLtX ? ivy : ivx);
                  tNew = (float)(edge - origin) * iv;
                }
                tnz = isZ ? tNew : tnz;
                tny = (!isZ && yLtX) ? tNew : tny;
                tnx = (!isZ && !yLtX) ? tNew : tnx;
              }
            }
            nWalk = __popcll(__ballot(state == ST_WALK));
          }
          XSTAMP(1);
          XCOUNT(8, 1); XCOUNT(9, __popcll(__ballot(state != ST_WALK && state != ST_DEAD)));
          // ---- lanes that stopped: runs of one-extinction layers, then hand finished legs over ----
          if (LAZY) {
            if (state == ST_JUMP) tryJump();
            if (lazy && state != ST_WALK && state != ST_DEAD) resolveXY();
          }
          const int ix = ex - (dx >= 0.0f ? 1 : 0), iy = ey - offY - (dy >= 0.0f ? 1 : 0), iz = ez - offZ - (dz >= 0.0f ? 1 : 0);
          if (state == ST_TOP) {  // out the top, computeRT :573-617: tally here, the slot goes to the dead ring
            const uint4 q3 = s_pool[5 * slot + 3];
            const unsigned long long dep = weight_to_fixed(__uint_as_float(q3.x));
            if (PRIV) atomicAdd(reinterpret_cast<unsigned long long *>(s_slab + (ix + p.nx * iy)), dep);
            else {
              const uint32_t batch = s_pool[5 * slot + 4].x;
              atomicAdd(reinterpret_cast<unsigned long long *>(p.slabs + (unsigned long long)batch * p.slabStride + (ix + p.nx * iy)), dep);
            }
          } else if (state == ST_COLLIDE || state == ST_SURFACE) {
            unsigned info;
            if (state == ST_COLLIDE) {  // opticalProperties.f95:1729-1738: the point where tau is used up
              const double s = (double)(tcur + div_fast(tau - acc, extCur));
              px = px + s * (double)dx;
              py = py + s * (double)dy;
              pz = pz + s * (double)dz;
              info = (unsigned)ix | ((unsigned)iy << 10) | ((unsigned)iz << 20) | (XK_COLLIDE << 30);
            } else {  // where the leg met the surface (:1809-1812)
              px = px + (double)tcur * (double)dx;
              py = py + (double)tcur * (double)dy;
              pz = p.zSurf;
              info = (unsigned)ix | ((unsigned)iy << 10) | (XK_SURFACE << 30);
            }
            s_pool[5 * slot] = make_uint4((unsigned)__double2loint(px), (unsigned)__double2hiint(px), (unsigned)__double2loint(py), (unsigned)__double2hiint(py));
            uint2 *h = reinterpret_cast<uint2 *>(s_pool + 5 * slot + 1);
            h[0] = make_uint2((unsigned)__double2loint(pz), (unsigned)__double2hiint(pz));
            uint2 *g = reinterpret_cast<uint2 *>(s_pool + 5 * slot + 2);
            g[1] = make_uint2(info, __float_as_uint(extCur));
          }
          xq_push(s_qctl + XQ_EVENT, qEvent, state == ST_COLLIDE || state == ST_SURFACE, (unsigned)slot, lane, laneBelow);
          xq_push(s_qctl + XQ_DEAD, qDead, state == ST_TOP, (unsigned)slot, lane, laneBelow);
          if (state == ST_COLLIDE || state == ST_SURFACE || state == ST_TOP) state = ST_DEAD;
        }
      }
      XSTAMP(2);
      const bool starving = __ballot(state != ST_DEAD) == 0ull && xq_count(s_qctl + XQ_WALK) <= 0;
      // ================= a dense pass over 64 collisions / surface hits ===================================
      const int cntE = xq_count(s_qctl + XQ_EVENT);
      if (cntE >= kWave || (starving && cntE > 0)) {
        const int es = xq_pop(s_qctl + XQ_EVENT, qEvent, true, starving ? 1 : kWave, lane, laneBelow);
        if (__ballot(es >= 0) != 0ull) {
          worked = true;
          XCOUNT(10, 1); XCOUNT(11, __popcll(__ballot(es >= 0)));
          bool alive = false;
          if (es >= 0) {
            const uint4 q0 = s_pool[5 * es], q1 = s_pool[5 * es + 1], q2 = s_pool[5 * es + 2], q3 = s_pool[5 * es + 3], q4 = s_pool[5 * es + 4];
            float edx = __uint_as_float(q1.z), edy = __uint_as_float(q1.w), edz = __uint_as_float(q2.x);
            const int eix = (int)(q2.z & 1023u), eiy = (int)((q2.z >> 10) & 1023u);
            int eiz = (int)((q2.z >> 20) & 1023u);
            const unsigned kind = q2.z >> 30;
            float w = __uint_as_float(q3.x);
            const uint32_t idLo = q3.y, idHi = q3.z;
            uint32_t event = q3.w;
            const uint32_t batch = q4.x;
            const float uX = __uint_as_float(q4.y), uY = __uint_as_float(q4.z), uZ = __uint_as_float(q4.w);
            const int ecell = eix + p.nx * (eiy + p.ny * eiz);
            uint4 o1 = q1;  // (pz may change: surface)
            alive = true;
            if (kind == XK_COLLIDE) {
              // scattering event, computeRT :703-821 (trace_kernel: same arithmetic)
              const long long nvox = (long long)ncol * p.nz;
              int c = 0;
              float ssa;
              int pfEntry;
              if (!gridLds && p.rec) {
                const uint4 rc = p.rec[ecell];
                c = uZ >= __uint_as_float(rc.x) ? 1 : 0;
                ssa = __uint_as_float(c ? rc.z : rc.y);
                pfEntry = (int)(c ? (rc.w >> 16) : (rc.w & 0xffffu));
              } else {
                const float *cumA = gridLds ? s_cum : p.cum;
                const float *ssaA = gridLds ? s_ssa : p.ssa;
                const uint16_t *pfiA = gridLds ? s_pfi : p.pfi;
                if (p.nc > 1) {
                  for (int k = 0; k < p.nc - 1; k++)
                    if (uZ >= cumA[(long long)k * nvox + ecell]) c = k + 1;
                }
                ssa = ssaA[(long long)c * nvox + ecell];
                pfEntry = pfiA[(long long)c * nvox + ecell];
              }
              if (ssa < 1.0f) {  // absorption :765-771
                const unsigned long long dep = weight_to_fixed(w * (1.0f - ssa));
                if (PRIV) atomicAdd(reinterpret_cast<unsigned long long *>(s_slab + 2 * ncol + ecell), dep);
                else atomicAdd(reinterpret_cast<unsigned long long *>(p.slabs + (unsigned long long)batch * p.slabStride + 2 * ncol + ecell), dep);
                w = w * ssa;
              }
              if (p.useRR && w < 0.5f) {  // Russian roulette :805-811
                uint32_t r1[4];
                philox4x32_10(event, 1u, idLo, idHi, p.seedLo, p.seedHi, r1);
                w = u01(r1[1]) >= w ? 0.0f : 1.0f;
              }
              if (w <= FLT_MIN) {
                alive = false;
              } else {
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
                float AX, AY;
                sincos_2pi(uY, AX, AY);
#ifdef MCBRAT_PRECISE_MATH
                float B = sqrtf(1.0f - cs * cs);
#else
                float B = __builtin_amdgcn_sqrtf(1.0f - cs * cs);
#endif
                AX = AX * B;
                AY = AY * B;
                B = edx * AX - edy * AY;
                const float D = cs - div_fast(B, 1.0f + fabsf(edz));
                const float ndx = edx * D + AX, ndy = edy * D - AY;
                edz = edz * cs - copysignf(fabsf(B), edz * B);
                edx = ndx; edy = ndy;
              }
            } else {
              // surface, computeRT :619-676 (Lambertian); fluxDown gets the incident weight :634
              eiz = 0;
              const unsigned long long dep = weight_to_fixed(w);
              if (PRIV) atomicAdd(reinterpret_cast<unsigned long long *>(s_slab + ncol + (eix + p.nx * eiy)), dep);
              else atomicAdd(reinterpret_cast<unsigned long long *>(p.slabs + (unsigned long long)batch * p.slabStride + ncol + (eix + p.nx * eiy)), dep);
              float mu = sqrtf(uX);
              if (!(fabsf(mu) > 2.0f * FLT_MIN)) {
                mu = sqrtf(uZ);
                uint32_t r[4];
                for (uint32_t j = 0; !(fabsf(mu) > 2.0f * FLT_MIN); j++) {
                  if ((j & 3u) == 0) philox4x32_10(event, 2u + (j >> 2), idLo, idHi, p.seedLo, p.seedHi, r);
                  mu = sqrtf(u01(pick4(r, j & 3u)));
                }
              }
              w = (float)((double)w * (double)p.albedo);  // :673
              if (w <= FLT_MIN) {
                alive = false;
              } else {
                const float sinTheta = sqrtf(1.0f - mu * mu);
                float cphi, sphi;
                sincos_2pi(uY, cphi, sphi);
                edx = sinTheta * cphi; edy = sinTheta * sphi; edz = mu;
              }
            }
            if (alive) {  // the next leg: tau and the Philox block that serves it
              event++;
              uint32_t r[4];
              philox4x32_10(event, 0u, idLo, idHi, p.seedLo, p.seedHi, r);
#ifdef MCBRAT_PRECISE_MATH
              const float ntau = -logf(fmaxf(FLT_MIN, u01(r[0])));
#else
              const float ntau = -0.693147182f * __builtin_amdgcn_logf(fmaxf(FLT_MIN, u01(r[0])));
#endif
              o1.z = __float_as_uint(edx); o1.w = __float_as_uint(edy);
              s_pool[5 * es + 1] = o1;
              s_pool[5 * es + 2] = make_uint4(__float_as_uint(edz), __float_as_uint(ntau),
                                             (unsigned)eix | ((unsigned)eiy << 10) | ((unsigned)eiz << 20), q2.w);
              s_pool[5 * es + 3] = make_uint4(__float_as_uint(w), idLo, idHi, event);
              s_pool[5 * es + 4] = make_uint4(batch, __float_as_uint(u01(r[1])), __float_as_uint(u01(r[2])), __float_as_uint(u01(r[3])));
            }
          }
          xq_push(s_qctl + XQ_WALK, qWalk, es >= 0 && alive, (unsigned)es, lane, laneBelow);
          xq_push(s_qctl + XQ_DEAD, qDead, es >= 0 && !alive, (unsigned)es, lane, laneBelow);
        }
      }
      XSTAMP(3);
      // ================= 64 dead slots: new photons (getNextPhoton + computeRT :466-508) ===================
      const int cntD = xq_count(s_qctl + XQ_DEAD);
      if (cntD >= kWave || (starving && cntD > 0 && cntE <= 0)) {
        const int ds = xq_pop(s_qctl + XQ_DEAD, qDead, true, starving ? 1 : kWave, lane, laneBelow);
        const unsigned long long mHave = __ballot(ds >= 0);
        if (mHave != 0ull) {
          worked = true;
          XCOUNT(12, 1); XCOUNT(13, __popcll(mHave));
          const int nHave = __popcll(mHave), rank = __popcll(mHave & laneBelow);
          unsigned long long myIdx;
          bool valid;
          uint32_t batch;
          if (PRIV) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&s_cursor[0], (unsigned)nHave);
            base = (unsigned)__shfl((int)base, 0);
            const unsigned k = base + (unsigned)rank;
            valid = k < unitCount;
            myIdx = unitFirst + k;
            batch = unitBatch;
          } else {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(p.counter, (unsigned long long)nHave);
            const uint32_t bl = __shfl((int)(uint32_t)base, 0), bh = __shfl((int)(uint32_t)(base >> 32), 0);
            myIdx = (((unsigned long long)bh << 32) | bl) + (unsigned long long)rank;
            valid = myIdx < p.total;
            batch = 0;
          }
          valid = valid && ds >= 0;
          if (valid) {
            if (!PRIV) batch = (uint32_t)(myIdx / p.ppb);
            const unsigned long long id = p.firstPhoton + myIdx;
            const uint32_t idLo = (uint32_t)id, idHi = (uint32_t)(id >> 32);
            uint32_t r[4];
            philox4x32_10(0u, 0u, idLo, idHi, p.seedLo, p.seedHi, r);
            // newPhotonStream_Directional, monteCarloIllumination.f95:88-96
            const double fx = (double)u01(r[0]), fy = (double)u01(r[1]);
            const double lx = p.x0 + fx * (p.xMax - p.x0), ly = p.y0 + fy * (p.yMax - p.y0);  // :480-482
            int jx, jy;
            if (p.xyRegular) {  // findXYIndicies :1558-1562
              jx = min((int)((lx - p.x0) * p.invDX), p.nx - 1);
              jy = min((int)((ly - p.y0) * p.invDY), p.ny - 1);
            } else {
              jx = find_cell(s_edge, p.nx, lx);
              jy = find_cell(s_edge + offY, p.ny, ly);
            }
            const int jz = p.izLaunch;
            const int lcell = jx + p.nx * (jy + p.ny * jz);
            const float lext = gridLds ? s_ext[lcell] : p.ext[lcell];
            philox4x32_10(1u, 0u, idLo, idHi, p.seedLo, p.seedHi, r);
#ifdef MCBRAT_PRECISE_MATH
            const float ntau = -logf(fmaxf(FLT_MIN, u01(r[0])));
#else
            const float ntau = -0.693147182f * __builtin_amdgcn_logf(fmaxf(FLT_MIN, u01(r[0])));
#endif
            s_pool[5 * ds] = make_uint4((unsigned)__double2loint(lx), (unsigned)__double2hiint(lx), (unsigned)__double2loint(ly), (unsigned)__double2hiint(ly));
            s_pool[5 * ds + 1] = make_uint4((unsigned)__double2loint(p.zLaunch), (unsigned)__double2hiint(p.zLaunch),
                                            __float_as_uint(p.dir0[0]), __float_as_uint(p.dir0[1]));
            s_pool[5 * ds + 2] = make_uint4(__float_as_uint(p.dir0[2]), __float_as_uint(ntau),
                                            (unsigned)jx | ((unsigned)jy << 10) | ((unsigned)jz << 20), __float_as_uint(lext));
            s_pool[5 * ds + 3] = make_uint4(__float_as_uint(1.0f), idLo, idHi, 1u);
            s_pool[5 * ds + 4] = make_uint4(batch, __float_as_uint(u01(r[1])), __float_as_uint(u01(r[2])), __float_as_uint(u01(r[3])));
          }
          xq_push(s_qctl + XQ_WALK, qWalk, valid, (unsigned)ds, lane, laneBelow);
          const int nRetire = __popcll(__ballot(ds >= 0 && !valid));  // no photons left: the slot goes out of use
          if (nRetire > 0 && lane == 0) atomicSub(&s_qctl[XQ_LIVE], (unsigned)nRetire);
        }
      }
      XSTAMP(4);
      if (x_load_uniform(&s_qctl[XQ_LIVE]) == 0u) break;  // every slot is out of use: nothing is in flight anywhere
      if (!worked) __builtin_amdgcn_s_sleep(8);
    }

    if (!PRIV) break;
    __syncthreads();
    XSTAMP(5);
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
#ifdef MCBRAT_XSTATS
  if (p.counters && lane == 0) for (int i = 0; i < 14; i++) atomicAdd(p.counters + 16 + i, xs[i]);
#endif
}


}  // namespace mcbrat
