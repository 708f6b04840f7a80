"""Regenerates the committed golden fixtures from the ORACLE (oracle/mcmc_oracle.c).

The reference itself cannot be built in this image (empty submodules), so these vectors
pin the oracle against regressions and give the GPU tests fixed inputs; the only vector
that comes from outside is the published PCG32 known-answer sequence in pcg32_kat.json.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mcmc_amd import covid_config as CC  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    # (ii) response table for the covid.ipynb DiseaseParams
    p = O.DiseaseParams(**CC.ipynb_disease_params())
    grid = O.traj_generate(p, 10_000_000)
    R = O.traj_response(grid)
    np.save(os.path.join(HERE, "R_ipynb_oracle.npy"), R)

    # (i) pcg32 streams.  kat = the published PCG32 demo sequence for pcg32(42, 54).
    streams = {}
    for st, sq in [(0, 0), (0, 1), (0, 7), (42, 54)]:
        r = O.Rng(st, sq); u = [r.next_uint() for _ in range(16)]
        r = O.Rng(st, sq); f = [float(r.next_float()) for _ in range(16)]
        r = O.Rng(st, sq); d = [r.next_double() for _ in range(16)]
        streams[f"{st},{sq}"] = {"uint": u, "float_hex": [float(x).hex() for x in f],
                                 "double_hex": [x.hex() for x in d]}
    r = O.Rng()
    streams["default"] = {"uint": [r.next_uint() for _ in range(16)]}
    kat = {"published_pcg32_42_54": [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e],
           "oracle_streams": streams,
           "note": "published_pcg32_42_54 is the known-answer output of the PCG32 reference demo "
                   "(pcg32_srandom(42,54)); oracle_streams are regression vectors of oracle/mcmc_oracle.c"}
    json.dump(kat, open(os.path.join(HERE, "pcg32_kat.json"), "w"), indent=1)

    # (iii)-(v) model vectors at 200 days: eval LL, prior, 40-step traces for 4 chains
    def fwd(cfg, x):
        return O.Model(cfg, R).eval(x, curves=True)[1][7]
    out = {}
    for contract in (0, 1):
        cfg = CC.synthetic_config(200, fwd, initial_beta_mild=0.8, initial_beta_high=2.5, contract=contract)
        m = O.Model(cfg, R, contract=contract)
        x0 = CC.initial_params(cfg)
        rng = O.Rng(0, 0)
        xs = np.array([[x0[k] + (np.float32(rng.next_float()) - np.float32(0.5)) * x0[k] * 0.05
                        for k in range(x0.size)] for _ in range(4)])
        tgt = O.make_target("temp", 1.0)
        rec = {"deaths_hex": [float(v).hex() for v in cfg["deaths"]], "chains": []}
        for c in range(4):
            ch = O.OracleChain(m, c + 1, xs[c])
            tr = {"ll0": [ch.ll[0].hex(), ch.ll[1].hex()], "accept": [], "delta_hex": [], "rng": []}
            for _ in range(40):
                a = ch.step(tgt, 0.0, True)
                tr["accept"].append(int(a)); tr["delta_hex"].append(float(ch.c.last_delta).hex())
                tr["rng"].append([int(ch.rng_state[0]), int(ch.rng_state[1])])
            tr["ll_end"] = [ch.ll[0].hex(), ch.ll[1].hex()]
            tr["x_end_sha"] = __import__("hashlib").sha256(ch.x.tobytes()).hexdigest()
            rec["chains"].append(tr)
        out[f"contract{contract}"] = rec
    json.dump(out, open(os.path.join(HERE, "covid200_traces.json"), "w"))
    print("golden fixtures written")


if __name__ == "__main__":
    main()
