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
//  * one photon per work-item, persistent 64-wide waves; a wave pulls chunks of photon ids
//    from one global counter and hands them to its idle lanes with ballot/popcount ranks
//    (regeneration: a lane whose photon left the domain, was absorbed by the surface or lost
//    at Russian roulette is refilled, so dead lanes never ride along);
//  * the loop is flattened: every iteration each walking lane advances exactly one voxel
//    face; launches / collisions / surface reflections are deferred until enough lanes of
//    the wave wait for them (eventThreshold), which keeps both halves of the loop dense;
//  * cell-authoritative voxel walk: per-axis parametric distance to the next face is kept in
//    registers and only the crossed axis is updated; the position is materialised once per
//    leg (statistically, not bitwise, identical to the reference's position-stepping walk);
//  * counter-based Philox4x32-10: key = seed, counter = (event, block, photon id).  Each role
//    has a fixed slot so that one block per leg serves (tau, component, roulette, angle):
//       event 0 (launch)   block 0 = [x, y, -, -]                      (solar source)
//                          block 0 = [select, r1, r2, r3], block 1.. (emission source)
//       event e >= 1       block 0 = [tau, A, B, C]
//            collision: component = A, roulette = B, angle = C,
//                       next_direct round k -> block 1 + k/2, elements 2(k&1), 2(k&1)+1
//            surface:   mu = sqrt(A) (retry C, then block 1..), phi = 2 pi B
//    (the oracle's Philox mode uses the same table);
//  * cell edges and, when they fit, the inverse phase-function tables are staged in LDS by
//    coalesced loads; extinction / ssa / phase index are float / float / u16 grids in HBM
//    that stay resident in L2 + Infinity Cache;
//  * tallies are signed 64-bit fixed point (2^-32) added with global atomics into one slab
//    per batch: integer sums are order-independent, so results are bitwise reproducible and
//    independent of the number of GPUs.  fluxAbsorbed is the column sum of the volume tally
//    (the reference adds the same deposit to both, :766-769).
#include <float.h>

#include "mcbrat_device.h"

namespace mcbrat {

constexpr int kBlock = 256;
constexpr unsigned long long kChunk = 256;  // photon ids a wave takes per global atomic

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// getRandomReal semantics (src/RandomNumbersForMC.f95:277-301): u32/(2^32-1) in double,
// rounded to float, both ends attainable.
__device__ __forceinline__ float u01(uint32_t u) { return (float)((double)u * (1.0 / 4294967295.0)); }

// element i (0..3) of a Philox block without dynamic register indexing
__device__ __forceinline__ uint32_t pick4(const uint32_t (&r)[4], uint32_t i) {
  const uint32_t a = (i & 1u) ? r[1] : r[0], b = (i & 1u) ? r[3] : r[2];
  return (i & 2u) ? b : a;
}

__device__ __forceinline__ void tally_add(long long *addr, double v) {
  atomicAdd(reinterpret_cast<unsigned long long *>(addr),
            static_cast<unsigned long long>(__double2ll_rn(v * kTallyScale)));
}

// findIndex(value, table) for cell edges: largest i with e[i] <= v, clamped to [0, n-1]
// (src/numericUtilities.f95:207-260, no first guess).
__device__ __forceinline__ int find_cell(const double *e, int n, double v) {
  int lo = 0, hi = n;  // cells lo..hi-1
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

template <bool TBL_LDS, bool DEBUG>
__global__ void __launch_bounds__(kBlock) trace_kernel(const DevParams p) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double *s_xe = reinterpret_cast<double *>(smem_raw);
  double *s_ye = s_xe + (p.nx + 1);
  double *s_ze = s_ye + (p.ny + 1);
  float *s_tbl = reinterpret_cast<float *>(s_ze + (p.nz + 1));
  {
    const int nEdges = p.nx + p.ny + p.nz + 3;
    for (int i = threadIdx.x; i < nEdges; i += kBlock) s_xe[i] = p.edges[i];
    if (TBL_LDS)
      for (int i = threadIdx.x; i < p.tblTotalFloats; i += kBlock) s_tbl[i] = p.tables[i];
    __syncthreads();
  }
  const float *__restrict__ tbl = TBL_LDS ? s_tbl : p.tables;

  const int lane = threadIdx.x & (kWave - 1);
  const unsigned long long laneBelow = (1ull << lane) - 1ull;
  const int ncol = p.nx * p.ny;
  const double zSurf = p.zSurf;

  // lane state ------------------------------------------------------------------
  int state = ST_DEAD;
  bool more = true;
  unsigned long long g = 0;  // photon index inside this launch
  uint32_t idLo = 0, idHi = 0, event = 0;
  double px = 0, py = 0, pz = 0;          // leg origin
  float dx = 0, dy = 0, dz = 1;           // direction cosines
  double ivx = 0, ivy = 0, ivz = 0;       // 1/direction
  double tnx = 0, tny = 0, tnz = 0, tcur = 0;  // distance along the leg to the next x/y/z face
  float acc = 0, tau = 0, w = 0, extCur = 0, uA = 0, uB = 0, uC = 0;
  int ix = 0, iy = 0, iz = 0, wx = 0, wy = 0;
  int nScat = 0, nLegs = 0;
  unsigned long long chunkNext = 0, chunkEnd = 0;  // wave-uniform
  // DEBUG counters (per lane, flushed at the end)
  unsigned int cLegs = 0, cCross = 0, cColl = 0, cAbs = 0, cTop = 0, cSurf = 0, cKill = 0, cSurv = 0;

  for (;;) {
    const int nWalk = __popcll(__ballot(state == ST_WALK));
    if (nWalk < p.eventThreshold) {  // wave-uniform
      bool needLeg = false;
      // ---- regeneration: hand photon ids to idle lanes --------------------------
      const unsigned long long want = __ballot(state == ST_DEAD && more);
      if (want != 0ull) {  // wave-uniform
        const int nWant = __popcll(want);
        const int rank = __popcll(want & laneBelow);
        unsigned long long myId = chunkNext + (unsigned long long)rank;
        const unsigned long long avail = chunkEnd - chunkNext;
        if (avail < (unsigned long long)nWant) {  // wave-uniform: take a new chunk
          unsigned long long base = 0;
          if (lane == 0) base = atomicAdd(p.counter, kChunk);
          const uint32_t bl = __shfl((int)(uint32_t)base, 0), bh = __shfl((int)(uint32_t)(base >> 32), 0);
          base = ((unsigned long long)bh << 32) | bl;
          if ((unsigned long long)rank >= avail) myId = base + ((unsigned long long)rank - avail);
          chunkNext = base + ((unsigned long long)nWant - avail);
          chunkEnd = base + kChunk;
        } else {
          chunkNext += (unsigned long long)nWant;
        }
        if (state == ST_DEAD && more) {
          if (myId < p.total) {
            // ---- launch: getNextPhoton + computeRT :466-508 ------------------------
            g = myId;
            const unsigned long long id = p.firstPhoton + g;
            idLo = (uint32_t)id; idHi = (uint32_t)(id >> 32);
            event = 0; nScat = 0; nLegs = 0;
            uint32_t r[4];
            philox4x32_10(0u, 0u, idLo, idHi, p.seedLo, p.seedHi, r);
            double fx, fy, fz;  // fractional launch position in [0,1]
            float mu = 0.f, phi = 0.f;
            bool fromSurface = false;
            if (p.srcKind == 0) {  // newPhotonStream_Directional, monteCarloIllumination.f95:88-96
              fx = (double)u01(r[0]);
              fy = (double)u01(r[1]);
              fz = 0.0;
              dx = p.dir0[0]; dy = p.dir0[1]; dz = p.dir0[2];
            } else {  // newPhotonStream_BBEmission :481-516
              const float sel = u01(r[0]);
              if ((double)sel > p.fracAtms) {  // surface emission :484-493
                fromSurface = true;
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
              ix = find_cell(s_xe, p.nx, px);
              iy = find_cell(s_ye, p.ny, py);
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
              pz = s_ze[iz] + (t - fl) * (s_ze[iz + 1] - s_ze[iz]);
            }
            if (p.lwFlag && pz > 0.0) {  // :504-508 emission counts as negative absorption
              long long *slab = p.slabs + (g / p.ppb) * p.slabStride;
              tally_add(slab + 2 * ncol + (ix + p.nx * (iy + p.ny * iz)), -1.0);
            }
            (void)fromSurface;
            needLeg = true;
          } else {
            more = false;
          }
        }
      }
      // ---- deferred events ------------------------------------------------------
      if (state == ST_COLLIDE) {
        // scattering event, computeRT :703-821 (zero-extinction back-step :728-754 cannot
        // arise: a collision is only declared inside a cell with extinction > 0)
        nScat++;
        if (DEBUG) cColl++;
        const int cell = ix + p.nx * (iy + p.ny * iz);
        const long long nvox = (long long)ncol * p.nz;
        int c = 0;  // component pick :759-760 (findIndex over [0, cumExt(:)])
        for (int k = 0; k < p.nc - 1; k++)
          if (uA >= p.cum[(long long)k * nvox + cell]) c = k + 1;
        const float ssa = p.ssa[(long long)c * nvox + cell];
        if (ssa < 1.0f) {  // absorption :765-771
          long long *slab = p.slabs + (g / p.ppb) * p.slabStride;
          tally_add(slab + 2 * ncol + cell, (double)w * (1.0 - (double)ssa));
          w = w * ssa;
          if (DEBUG) cAbs++;
        }
        if (p.useRR && w < 0.5f) {  // Russian roulette :805-811, RussianRouletteW = 1
          if (uB >= w) { w = 0.0f; if (DEBUG) cKill++; }
          else { w = 1.0f; if (DEBUG) cSurv++; }
        }
        if (w <= FLT_MIN) {  // :812
          if (DEBUG && p.fates) p.fates[g] = mcbrat_fate{2, ix + 1, iy + 1, iz + 1, nScat, nLegs, 0.0f};
          state = ST_DEAD;
        } else {
          // computeScatteringAngle :1594-1621 (table point count N, floor-type lookup as written)
          const int pf = p.pfi[(long long)c * nvox + cell];
          const int n = p.tblNSteps[c];
          const float *t = tbl + p.tblOffset[c] + (long long)pf * n;
          const int ai = (int)(uC * (float)n) + 1;
          float ang;
          if (ai < n) {
            const float left = uC - (float)(ai - 1) / (float)n;
            ang = (1.0f - left) * t[ai - 1] + left * t[ai];
          } else {
            ang = t[n - 1];
          }
          const float cs = cosf(ang);
          // next_direct :1921-1948
          float AX = 0.f, AY = 0.f, D = 2.0f;
          uint32_t r[4];
          for (uint32_t k = 0; D > 1.0f; k++) {
            if ((k & 1u) == 0) philox4x32_10(event, 1u + (k >> 1), idLo, idHi, p.seedLo, p.seedHi, r);
            AX = 1.0f - 2.0f * u01((k & 1u) ? r[2] : r[0]);
            AY = 1.0f - 2.0f * u01((k & 1u) ? r[3] : r[1]);
            D = AX * AX + AY * AY;
          }
          float B = sqrtf((1.0f - cs * cs) / D);
          AX = AX * B;
          AY = AY * B;
          B = dx * AX - dy * AY;
          D = cs - B / (1.0f + fabsf(dz));
          dx = dx * D + AX;
          dy = dy * D - AY;
          dz = dz * cs - copysignf(fabsf(B), dz * B);
          needLeg = true;
        }
      } else if (state == ST_SURFACE) {
        // surface, computeRT :619-676 (Lambertian)
        long long *slab = p.slabs + (g / p.ppb) * p.slabStride;
        tally_add(slab + ncol + (ix + p.nx * iy), (double)w);  // fluxDown gets the incident weight :634
        nScat++;
        if (DEBUG) cSurf++;
        float mu = sqrtf(uA);
        if (!(fabsf(mu) > 2.0f * FLT_MIN)) {
          mu = sqrtf(uC);
          uint32_t r[4];
          for (uint32_t j = 0; !(fabsf(mu) > 2.0f * FLT_MIN); j++) {
            if ((j & 3u) == 0) philox4x32_10(event, 1u + (j >> 2), idLo, idHi, p.seedLo, p.seedHi, r);
            mu = sqrtf(u01(pick4(r, j & 3u)));
          }
        }
        const float phi = (2.0f * 3.14159274f) * uB;
        const float wIn = w;
        w = (float)((double)w * (double)p.albedo);  // :673
        if (w <= FLT_MIN) {
          if (DEBUG && p.fates) p.fates[g] = mcbrat_fate{1, ix + 1, iy + 1, 1, nScat, nLegs, wIn};
          state = ST_DEAD;
        } else {
          const float sinTheta = sqrtf(1.0f - mu * mu);
          dx = sinTheta * cosf(phi); dy = sinTheta * sinf(phi); dz = mu;
          needLeg = true;
        }
      }
      // ---- start the next leg: tau and the per-axis face distances -------------------
      if (needLeg) {
        event++;
        nLegs++;
        if (DEBUG) cLegs++;
        uint32_t r[4];
        philox4x32_10(event, 0u, idLo, idHi, p.seedLo, p.seedHi, r);
        const float u = u01(r[0]);
        tau = -logf(fmaxf(FLT_MIN, u));  // :554
        uA = u01(r[1]); uB = u01(r[2]); uC = u01(r[3]);
        acc = 0.0f; tcur = 0.0; wx = 0; wy = 0;
        // opticalProperties.f95:1690-1712: side 1 where direction >= 0
        if (fabsf(dx) >= 2.0f * FLT_MIN) { ivx = 1.0 / (double)dx; tnx = (s_xe[ix + (dx >= 0.0f ? 1 : 0)] - px) * ivx; }
        else { ivx = 0.0; tnx = DBL_MAX; }
        if (fabsf(dy) >= 2.0f * FLT_MIN) { ivy = 1.0 / (double)dy; tny = (s_ye[iy + (dy >= 0.0f ? 1 : 0)] - py) * ivy; }
        else { ivy = 0.0; tny = DBL_MAX; }
        if (fabsf(dz) >= 2.0f * FLT_MIN) { ivz = 1.0 / (double)dz; tnz = (s_ze[iz + (dz >= 0.0f ? 1 : 0)] - pz) * ivz; }
        else { ivz = 0.0; tnz = DBL_MAX; }
        extCur = p.ext[ix + p.nx * (iy + p.ny * iz)];
        state = ST_WALK;
      }
      if (__ballot(state != ST_DEAD) == 0ull) break;  // wave-uniform: nothing alive, nothing left
    }

    // ---- one face per iteration: accumulateExtinctionAlongPath :1697-1814 --------------
    if (state == ST_WALK) {
      double tmin = tnx;
      int ax = 0;
      if (tny < tmin) { tmin = tny; ax = 1; }
      if (tnz < tmin) { tmin = tnz; ax = 2; }
      const double dtau = (tmin - tcur) * (double)extCur;
      if ((double)acc + dtau > (double)tau) {
        // :1729-1738 stop inside this cell
        const double s = tcur + (double)(tau - acc) / (double)extCur;
        px = px + s * (double)dx - (double)wx * p.Lx;
        py = py + s * (double)dy - (double)wy * p.Ly;
        pz = pz + s * (double)dz;
        state = ST_COLLIDE;
      } else {
        acc = (float)((double)acc + dtau);  // :1743
        tcur = tmin;
        if (DEBUG) cCross++;
        if (ax == 0) {
          if (dx >= 0.0f) { if (++ix == p.nx) { ix = 0; wx++; } }
          else { if (--ix < 0) { ix = p.nx - 1; wx--; } }  // periodic :1782-1788
          tnx = (s_xe[ix + (dx >= 0.0f ? 1 : 0)] + (double)wx * p.Lx - px) * ivx;
        } else if (ax == 1) {
          if (dy >= 0.0f) { if (++iy == p.ny) { iy = 0; wy++; } }
          else { if (--iy < 0) { iy = p.ny - 1; wy--; } }
          tny = (s_ye[iy + (dy >= 0.0f ? 1 : 0)] + (double)wy * p.Ly - py) * ivy;
        } else {
          iz += (dz >= 0.0f) ? 1 : -1;
          if (iz >= p.nz) {  // out the top :1801-1804, computeRT :573-617
            long long *slab = p.slabs + (g / p.ppb) * p.slabStride;
            tally_add(slab + (ix + p.nx * iy), (double)w);
            if (DEBUG) { cTop++; if (p.fates) p.fates[g] = mcbrat_fate{0, ix + 1, iy + 1, p.nz + 1, nScat, nLegs, w}; }
            state = ST_DEAD;
          } else if (iz < 0) {  // hit the bottom :1809-1812, computeRT :619-633
            px = px + tmin * (double)dx - (double)wx * p.Lx;
            py = py + tmin * (double)dy - (double)wy * p.Ly;
            pz = zSurf;
            iz = 0;
            state = ST_SURFACE;
          } else {
            tnz = (s_ze[iz + (dz >= 0.0f ? 1 : 0)] - pz) * ivz;
          }
        }
        if (state == ST_WALK) extCur = p.ext[ix + p.nx * (iy + p.ny * iz)];
      }
    }
  }

  if (DEBUG && p.counters) {
    atomicAdd(p.counters + 0, (unsigned long long)cLegs);
    atomicAdd(p.counters + 1, (unsigned long long)cCross);
    atomicAdd(p.counters + 2, (unsigned long long)cColl);
    atomicAdd(p.counters + 3, (unsigned long long)cAbs);
    atomicAdd(p.counters + 4, (unsigned long long)cTop);
    atomicAdd(p.counters + 5, (unsigned long long)cSurf);
    atomicAdd(p.counters + 6, (unsigned long long)cKill);
    atomicAdd(p.counters + 7, (unsigned long long)cSurv);
  }
}

// ---------------------------------------------------------------------------------------
// Per-batch epilogue on the device: normalisation (computeRadiativeTransfer :328-364),
// reportResults (:877-884, :966) and the driver's moments (monteCarloDriver.f95:1023-1050).
// Batches are folded in index order by one owner thread per output element, so the
// double sums are reproducible.
// ---------------------------------------------------------------------------------------
struct FinishParams {
  int nx, ny, nz, nBatches, xyRegular;
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

__global__ void finish_columns(const FinishParams f) {
  const int ncol = f.nx * f.ny;
  const long long M = 3 + 3LL * ncol + f.nz + (long long)ncol * f.nz;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncol) return;
  double s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0};
  float lastv[3] = {0, 0, 0};
  for (int b = 0; b < f.nBatches; b++) {
    const unsigned long long n = batch_photons(f, b);
    const float nppc = photons_per_column(f, col, n);
    const long long *slab = f.slabs + (unsigned long long)b * f.slabStride;
    long long rawAbs = 0;
    for (int k = 0; k < f.nz; k++) rawAbs += slab[2 * ncol + col + (long long)ncol * k];
    const long long raw[3] = {slab[col], slab[ncol + col], rawAbs};
    for (int q = 0; q < 3; q++) {
      const float v = (float)((double)raw[q] * kTallyInv) / nppc;  // :348-350
      f.colVals[((long long)b * 3 + q) * ncol + col] = v;
      s1[q] += (double)v * (double)(long long)n;
      s2[q] += (double)(long long)n * ((double)v * (double)v);
      lastv[q] = v;
    }
  }
  double *S1 = f.moments + 8, *S2 = f.moments + 8 + M;
  for (int q = 0; q < 3; q++) {
    S1[3 + (long long)q * ncol + col] += s1[q];
    S2[3 + (long long)q * ncol + col] += s2[q];
    f.last[3 + (long long)q * ncol + col] = lastv[q];
  }
}

__global__ void finish_volume(const FinishParams f) {
  const int ncol = f.nx * f.ny;
  const long long nvox = (long long)ncol * f.nz;
  const long long M = 3 + 3LL * ncol + f.nz + nvox;
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
  const long long M = 3 + 3LL * ncol + f.nz + (long long)ncol * f.nz;
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
