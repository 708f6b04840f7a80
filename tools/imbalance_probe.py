#!/usr/bin/env python3
"""Where do the slow waves of a launch run?  Needs the -DMCMC_STAMPS library:
   MCMC_HIP_LIB=.../libmcmc_hip_stamps.so python tools/imbalance_probe.py [chains]
Prints the distribution of per-wave launch cycles at equal work per tile, grouped by XCD, by SIMD and by
launch, and how much of a wave's deviation persists from one launch to the next."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
tiles = n // 64
cfg, R, xs, s = bench.build_workload(0, n, 1, 1)
s.run(0, 0, 25, 1000000)
s.step(16, 1.0, True)
logs = []
for rep in range(4):
    s.phase_cycles(reset=True)
    s.step(16, 1.0, True)
    cyc, hw, ech, eda, ehu = s.wave_log(min(tiles, 8192))
    logs.append((cyc, hw))
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
    hwid = hw.astype(np.int64) & 0xffffffff
    simd = (hwid >> 4) & 3
    cu = (hwid >> 8) & 0xf
    se = (hwid >> 13) & 7
    rel = cyc / cyc.mean()
    print(f"launch {rep}: mean {cyc.mean() / 16:.0f} cycles/step, sd {100 * rel.std():.2f} %, max {100 * (rel.max() - 1):+.2f} %, min {100 * (rel.min() - 1):+.2f} %")
    print("  by XCD :", " ".join(f"{x}:{100 * (rel[xcc == x].mean() - 1):+.2f}" for x in np.unique(xcc)))
    print("  by SIMD:", " ".join(f"{x}:{100 * (rel[simd == x].mean() - 1):+.2f}" for x in np.unique(simd)))
    print("  by SE  :", " ".join(f"{x}:{100 * (rel[se == x].mean() - 1):+.2f}" for x in np.unique(se)))
    print("  by tile mod 4 (wave of the workgroup):", " ".join(f"{x}:{100 * (rel[np.arange(len(rel)) % 4 == x].mean() - 1):+.2f}" for x in range(4)))
    print(f"  e-chunks per tile per step: mean {ech.mean() / 16:.1f} sd {ech.std() / 16:.1f} max {ech.max() / 16:.1f}; corr(cycles, e-chunks) {np.corrcoef(cyc, ech)[0, 1]:.3f}")
    print(f"  days with a correction / hull days: {eda.sum() / max(ehu.sum(), 1):.3f} (mean hull {ehu.mean() / 32:.1f} days per pass)")
    res = rel - np.array([rel[xcc == x].mean() for x in range(8)])[xcc]
    A = np.vstack([ech / ech.mean() - 1, np.ones_like(ech)]).T
    coef = np.linalg.lstsq(A, res, rcond=None)[0]
    print(f"  after removing XCD means: sd {100 * res.std():.2f} %; after also regressing on e-chunks: sd {100 * (res - A @ coef).std():.2f} % (slope {coef[0]:.3f})")
    d = s.params()[:, -1].reshape(tiles, 64).max(axis=1)
    print(f"  per-tile max delay: sd {d.std():.3f} days (equal work per tile when ~0)")
a, b = logs[-2][0], logs[-1][0]
print("correlation of a tile's deviation between consecutive launches:", np.corrcoef(a / a.mean(), b / b.mean())[0, 1])
same = logs[-2][1] == logs[-1][1]
print("tiles on the same SIMD in both launches:", same.mean())
