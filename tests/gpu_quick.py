"""Quick manual GPU sanity script (not collected by pytest): eval/derive/step parity."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from mcmc_amd import covid_config as CC, hip as H

R11 = np.load(os.path.join(os.path.dirname(__file__), "golden", "R_ipynb_oracle.npy"))
ORDER = [3, 4, 0, 2, 5, 7, 6, 8, 9, 10]   # oracle curve index of each of the 10 ABI curves
R10 = R11[ORDER]

def fwd(cfg, x):
    return O.Model(cfg, R11).eval(x, curves=True)[1][7]

def main():
    nd = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    nch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    for contract in (0, 1):
        cfg = CC.synthetic_config(nd, fwd, initial_beta_mild=0.8, initial_beta_high=2.5, contract=contract)
        m = O.Model(cfg, R11, contract=contract)
        x0 = CC.initial_params(cfg)
        s = H.CovidSampler(cfg, R10)
        xs = H.perturb_ics(x0, nch, 0.05)
        t = time.time(); ll = s.eval(xs); print("gpu eval s", time.time() - t)
        ref = np.array([[m.eval(x), m.prior(x)] for x in xs[:64]])
        print("contract", contract, "eval bitexact:", np.array_equal(ll[:64], ref), "maxrel", np.max(np.abs(ll[:64] - ref) / np.abs(ref)))
        cur = s.derive(xs[:3])
        refc = np.array([m.eval(x, curves=True)[1] for x in xs[:3]])
        print(" derive bitexact:", np.array_equal(cur, refc), "maxabs", np.nanmax(np.abs(cur - refc)))
        s.set_chains(xs, first_chain_id=1)
        print(" init ll bitexact:", np.array_equal(s.loglike()[:64], ref))
        nsteps = 8
        t = time.time(); s.step(nsteps, 1.0, True); dt = time.time() - t
        print(" %d steps x %d chains: %.3f s  -> %.1f steps/s" % (nsteps, nch, dt, nsteps * nch / dt))
        tgt = O.make_target("temp", 1.0)
        ok = True
        P = s.params(); LL = s.loglike(); RNG = s.rng_state(); ACC = s.accept_counts()
        for c in range(min(nch, 8)):
            oc = O.OracleChain(m, c + 1, xs[c])
            for _ in range(nsteps):
                oc.step(tgt, 0.0, True)
            same = (np.array_equal(oc.x, P[c]) and oc.ll == tuple(LL[c]) and oc.rng_state == tuple(int(v) for v in RNG[c])
                    and oc.c.n_accept == ACC[c])
            if not same:
                print("  chain", c, "MISMATCH", np.max(np.abs(oc.x - P[c])), oc.ll, LL[c], oc.rng_state, RNG[c], oc.c.n_accept, ACC[c])
            ok &= same
        print(" step parity (8 chains):", ok)
        s.close()

main()
