"""Development sweep of the ray-buffer parameters (radiance on the 128x128x64 cloud field, 4 directions)."""
import os, subprocess, sys, itertools
here = os.path.dirname(os.path.abspath(__file__))
for rr, nb in ((1, 20), (0, 10)):
    for short, iters, at in ((8, 32, 40), (4, 32, 40), (16, 32, 40), (8, 16, 40), (8, 64, 40), (8, 32, 56), (8, 32, 24), (4, 16, 56), (4, 64, 56)):
        env = dict(os.environ, MCBRAT_RAY_SHORT=str(short), MCBRAT_RAY_PASS_ITERS=str(iters), MCBRAT_RAY_PASS_AT=str(at))
        r = subprocess.run([sys.executable, os.path.join(here, "ray_defer_check.py"), "child", "landsat128", "4", str(rr), str(nb), "/tmp/raytune.npy"],
                           env=env, capture_output=True, text=True, timeout=600)
        print("roulette=%d short=%d passIters=%d passAt=%d: %s" % (rr, short, iters, at, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]), flush=True)
