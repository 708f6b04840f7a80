#!/usr/bin/env python3
"""Do two builds of the library give the same bits?  Runs bench.py's workload (small) for a few steps under each
library in a child process and compares parameters, log-likelihoods and RNG states.
usage: ab_equal.py libA.so libB.so [precision=f32_model] [chains=4096] [steps=24]"""
import hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, %r)
import numpy as np
import bench
prec, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg, R, xs, s = bench.build_workload(0, n, 1, 1, precision=prec)
s.run(0, 0, 3, 1000000)
s.step(steps, 1.0, True)
h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
print(json.dumps({"params": h(s.params()), "loglike": h(s.loglike()), "rng": h(s.rng_state()),
                  "accepted": int(s.accept_counts().sum()), "ll_mean": float(s.loglike().sum(axis=1).mean())}))
s.close()
""" % ROOT


def main():
    a, b = sys.argv[1], sys.argv[2]
    prec = sys.argv[3] if len(sys.argv) > 3 else "f32_model"
    n = sys.argv[4] if len(sys.argv) > 4 else "4096"
    steps = sys.argv[5] if len(sys.argv) > 5 else "24"
    res = []
    for lib in (a, b):
        env = dict(os.environ, MCMC_HIP_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", CHILD, prec, n, steps], env=env, capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            print(out.stderr[-2000:])
            raise SystemExit(f"{lib}: child failed ({out.returncode})")
        res.append(json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1]))
        print(lib, res[-1], flush=True)
    same = res[0] == res[1]
    print("IDENTICAL" if same else "DIFFERENT")
    raise SystemExit(0 if same else 1)


if __name__ == "__main__":
    main()
