import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spicey_amd import synth
from oracle.pyoracle import OracleBackend
from spicey_amd.lib import HipBackend
for kind in ['rc_ladder', 'diode_chain']:
    flat, dt, steps, src = synth.chain_batch(kind, 1000, [1], tran='.tran 1e-6 2e-5')
    ref = OracleBackend().run(flat, steps, dt, src)
    for T in [256, 512, 1024]:
        be = HipBackend(threads=T, interpreter=2)
        r = be.run(flat, steps, dt, src)
        ev = np.abs(r['out_v'] - ref['out_v']) / (1e-9 * np.abs(ref['out_v']) + 1e-12)
        ei = np.abs(r['out_i'] - ref['out_i']) / (1e-9 * np.abs(ref['out_i']) + 1e-12)
        bad = np.argwhere(ei > 1)
        print(kind, T, r['status'], ev.max(), ei.max(), len(bad), bad[:6].tolist(), [(r['out_i'][tuple(b)], ref['out_i'][tuple(b)]) for b in bad[:3]])
