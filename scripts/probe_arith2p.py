"""Probe of the two-phase range-coder encoder: the cases of tests/test_arith.py's two-phase test against the oracle, one report line per failure, and the
wall time of the call in its three forms (everything two-phase / default / one pass).  GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from htslib_amd import _native as nat
from tests import refutil
from tests.test_arith import two_phase_cases

eng = nat.Engine(0)
orc = refutil.ArithOracle()
datas, flags = two_phase_cases()
res = {}
for label, env in (("all two-phase", {"HG_ARITH_2P_MIN": "1"}), ("default", {}), ("one pass", {"HG_ARITH_2P": "0"})):
    for k in ("HG_ARITH_2P_MIN", "HG_ARITH_2P"): os.environ.pop(k, None)
    os.environ.update(env)
    eng.arith_encode_host(datas[:4], flags[:4])
    t = time.perf_counter()
    res[label] = eng.arith_encode_host(datas, flags)
    print("%-14s encode call %.1f ms" % (label, (time.perf_counter() - t) * 1e3), flush=True)
for k in ("HG_ARITH_2P_MIN", "HG_ARITH_2P"): os.environ.pop(k, None)
nbad = 0
for i, (d, fl) in enumerate(zip(datas, flags)):
    ref = orc.encode(d, fl)
    for label, enc in res.items():
        if enc[i] != ref:
            nbad += 1
            if nbad <= 40: print("n=%d distinct=%d flags=%d %s: DIFFERENT (len %d vs %d, first diff at %d)" % (len(d), len(set(d)), fl, label, len(enc[i]), len(ref),
                                 next((j for j in range(min(len(enc[i]), len(ref))) if enc[i][j] != ref[j]), -1)), flush=True)
outs, st = eng.cram_uncompress_blocks([(6, e, len(d)) for d, e in zip(datas, res["all two-phase"])])
nd = sum(1 for d, o, s in zip(datas, outs, st) if s != 0 or o != d)
print("done: %d streams, %d encoder mismatches, %d decode failures" % (len(datas), nbad, nd))
