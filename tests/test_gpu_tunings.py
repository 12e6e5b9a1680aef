"""Scheduling must not show in the results: workgroup size, thresholds of the queues, batches in flight and workgroups per
CU -- and, for the face-by-face walk, private or global tallies and dense or bricked grids -- only change WHEN a lane does
its work.  On random small domains (tests/test_gpu_parity.py::random_oracle_case) a random tuning must reproduce the
default run of the same walk bit for bit: every photon's fate and the moment array of a three-batch run.  Three walks:
face by face, layer-skipping walk with the clear-air flight, block walk."""
import os

import numpy as np
import pytest

from tests.test_gpu_parity import SEED, random_oracle_case
from tests import cases

pytestmark = pytest.mark.gpu

@pytest.fixture(scope="module")
def M():
    import mcbrat3d_amd
    return mcbrat3d_amd


FUZZ = int(os.environ.get("MCBRAT_FLIGHT_FUZZ", "12"))  # seeds of the random differential test (raise it for a soak run)


OPTIONS = ("jumpThreshold", "crossThreshold")  # by name (mcbrat_set_option), the rest through setTuning


def _run(M, case, mu0, phi0, rr, walk, tuning, n=12000):
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=rr)
    both = {**walk, **tuning}
    integ.setTuning(**{k: v for k, v in both.items() if k not in OPTIONS})
    integ.setOption(**{k: v for k, v in both.items() if k in OPTIONS})
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 9)
    fates = integ.traceFates(dom, new_RandomNumberSequence(SEED), photons, n)
    integ.resetMoments()
    integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n // 3, 3)
    mom = integ.moments().copy()
    integ.finalize()
    return fates, mom


@pytest.mark.timeout(120, method="thread")
@pytest.mark.parametrize("seed", range(FUZZ))
def test_random_tunings_give_the_same_histories(M, seed):
    case, mu0, phi0, rr = random_oracle_case(seed)
    rng = np.random.default_rng(77000 + seed)
    sched = dict(eventThreshold=int(rng.choice([1, 4, 16, 40, 64])), blockSize=int(rng.choice([0, 256, 512])),
                 launchThreshold=int(rng.choice([1, 8, 32])), surfaceThreshold=int(rng.choice([1, 12, 32])),
                 blocksPerCU=int(rng.choice([0, 1, 3])), maxBatchesInFlight=int(rng.choice([0, 1, 2])))
    # privateTallies 4 / 5: the wide plan -- one workgroup of 1024 lanes per compute unit with the tallies in its LDS, the
    # optical grid beside them (4) or left in global memory (5: for the block walk, extinction per block in LDS) -- which
    # the library chooses by itself only for tally slabs too large to share a compute unit (config 4's 70 KB)
    queues = dict(jumpThreshold=int(rng.choice([1, 8, 24])), crossThreshold=int(rng.choice([1, 8, 24])))  # (by name: mcbrat_set_option)
    layout = dict(privateTallies=int(rng.choice([0, 1, 4, 5, 6])), brickLayout=int(rng.integers(0, 2)))
    if layout["privateTallies"] >= 4:
        layout.update(blockSize=0, brickLayout=0)
    for name, walk, base_tuning, tuning in (
            ("face by face", dict(layerSkip=0, blockWalk=0), dict(eventThreshold=16, privateTallies=0, brickLayout=0), {**sched, **layout}),
            ("layers + flight", dict(layerSkip=3, blockWalk=0, privateTallies=0, brickLayout=0), dict(eventThreshold=16), {**sched, **queues}),
            # (layerSkip = 2: the layers without the clear-air flight, which the wide plan does not carry; privateTallies = 5:
            # with the grid in LDS too -- 4 -- the walk is the face-by-face one, the first row's)
            ("layers, wide plan", dict(layerSkip=2, blockWalk=0, brickLayout=0), dict(eventThreshold=16, privateTallies=0),
             {**{k: v for k, v in sched.items() if k != "blockSize"}, "privateTallies": 5}),
            ("block walk", dict(blockWalk=2), dict(eventThreshold=16),
             {**{k: v for k, v in sched.items() if k != "blockSize"}, **queues, "privateTallies": int(rng.choice([1, 4, 5, 6]))})):
        base = _run(M, case, mu0, phi0, rr, walk, base_tuning)
        got = _run(M, case, mu0, phi0, rr, walk, tuning)
        for f in base[0].dtype.names:
            assert np.array_equal(got[0][f], base[0][f], equal_nan=True), (name, f, tuning)
        assert np.array_equal(got[1], base[1], equal_nan=True), (name, tuning)


def _run_thermal(M, case, rr, walk, tuning, n=12000):
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    dom = cases.product_domain(case)
    nx, ny, nz = dom.numX, dom.numY, dom.numZ
    w = M.new_Weights(nx, ny, nz)
    M.emission_weighting(dom, w, case["sfc_temp"])
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=rr, LW_flag=1.0)
    integ.setTuning(**{**walk, **tuning})
    photons = M.new_PhotonStream(theseWeights=w, numberOfPhotons=10 ** 9)
    fates = integ.traceFates(dom, new_RandomNumberSequence(SEED), photons, n)
    integ.resetMoments()
    integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n // 3, 3)
    mom = integ.moments().copy()
    integ.finalize()
    return fates, mom


@pytest.mark.timeout(120, method="thread")
@pytest.mark.parametrize("seed", range(FUZZ))
def test_random_tunings_give_the_same_histories_thermal_source(M, seed):
    """The thermal source has defaults of its own (launch threshold 32, event threshold 8 for calls too small for the trial
    launches: its launch is the expensive part of a short history).  The library's defaults, the
    solar source's and a random choice must give every photon the same history and the same moment array."""
    case, _, _, rr = random_oracle_case(seed)
    rng = np.random.default_rng(78000 + seed)
    nx, ny, nz = len(case["xe"]) - 1, len(case["ye"]) - 1, len(case["ze"]) - 1
    case["temps"] = rng.uniform(230.0, 300.0, (nx, ny, nz))
    case["sfc_temp"] = float(rng.uniform(250.0, 320.0))
    case["lambda_um"] = float(rng.uniform(6.0, 14.0))
    sched = dict(eventThreshold=int(rng.choice([1, 4, 16, 40, 64])), launchThreshold=int(rng.choice([1, 8, 48])),
                 surfaceThreshold=int(rng.choice([1, 12, 32])), maxBatchesInFlight=int(rng.choice([0, 1, 2])))
    wide = dict(privateTallies=int(rng.choice([4, 5, 6])))  # the wide plan (config 4's), see above
    for name, walk, plans in (("face by face", dict(layerSkip=0, blockWalk=0, privateTallies=0), [wide]),
                              ("layers + flight", dict(layerSkip=3, blockWalk=0, privateTallies=0), []),  # (the wide plan carries no flight)
                              # (privateTallies = 5: with the grid in LDS too -- 4 -- the walk is the face-by-face one)
                              ("layers", dict(layerSkip=2, blockWalk=0, privateTallies=0), [dict(privateTallies=5)]),
                              ("block walk", dict(blockWalk=2), [wide])):
        base = _run_thermal(M, case, rr, walk, {})                                                # the thermal source's own defaults
        for tuning in [dict(eventThreshold=16, launchThreshold=8), sched] + [{**sched, **w} for w in plans]:  # the solar source's; a random choice; the same in the wide plan
            got = _run_thermal(M, case, rr, walk, tuning)
            for f in base[0].dtype.names:
                assert np.array_equal(got[0][f], base[0][f], equal_nan=True), (name, f, tuning)
            assert np.array_equal(got[1], base[1], equal_nan=True), (name, tuning)


@pytest.mark.timeout(120, method="thread")
@pytest.mark.parametrize("seed", range(FUZZ))
def test_random_call_sequences_async_equals_sync(M, seed):
    """A random sequence of calls -- computeRadiativeTransfer with random batch sizes and batch counts (the buffers of the
    context's streams grow in between), resetMoments, reads of the moments and of the last batch's results -- gives
    bit for bit the same in asynchronous mode (calls overlap on the GPU) as in synchronous mode."""
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    case, mu0, phi0, rr = random_oracle_case(seed)
    rng = np.random.default_rng(88000 + seed)
    ops = []
    for _ in range(int(rng.integers(4, 14))):
        r = rng.random()
        if r < 0.65:
            ops.append(("trace", int(rng.integers(1, 6000)), int(rng.integers(1, 6))))
        elif r < 0.8:
            ops.append(("reset",))
        elif r < 0.9:
            ops.append(("moments",))
        else:
            ops.append(("report",))
    ops.append(("trace", 2000, 2))
    out = {}
    for mode in ("sync", "async"):
        dom = cases.product_domain(case)
        integ = M.new_Integrator(dom)
        integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=rr)
        integ.setTuning(eventThreshold=16)
        integ.setAsync(mode == "async")
        photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 9)
        r = new_RandomNumberSequence(SEED)
        seen = []
        integ.resetMoments()
        for op in ops:
            if op[0] == "trace":
                assert integ.computeRadiativeTransfer(dom, r, photons, op[1], op[2]) == op[1] * op[2]
            elif op[0] == "reset":
                integ.resetMoments()
            elif op[0] == "moments":
                seen.append(integ.moments().copy())
            elif integ.moments()[0] > 0 or any(o[0] == "trace" for o in ops[:ops.index(op)]):
                res = integ.reportResults()
                seen.append(np.concatenate([np.ravel(np.asarray(res[k], np.float64)) for k in sorted(res)]))
        integ.synchronize()
        seen.append(integ.moments().copy())
        res = integ.reportResults()
        seen.append(np.concatenate([np.ravel(np.asarray(res[k], np.float64)) for k in sorted(res)]))
        integ.finalize()
        out[mode] = seen
    assert len(out["sync"]) == len(out["async"])
    for a, b in zip(out["sync"], out["async"]):
        assert np.array_equal(a, b, equal_nan=True), ops
