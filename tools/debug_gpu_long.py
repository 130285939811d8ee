import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spicey_amd import synth
from spicey_amd.lib import HipBackend
flat, dt, steps, src = synth.chain_batch('rc_ladder', 1000, [1], tran='.tran 1e-6 1e-2')
# small circuits first, like the test-suite does (different handles / allocations before the long run)
f2, dt2, st2, src2 = synth.chain_batch('diode_chain', 40, range(1, 8), tran='.tran 1e-6 3e-5')
first = None
for rep in range(8):
    HipBackend(threads=64, interpreter=2).run(f2, st2, dt2, src2)
    be = HipBackend(interpreter=2)
    r = be.run(flat, steps, dt, src)
    oi, ov = r['out_i'][0], r['out_v'][0]
    if first is None:
        first = (oi.copy(), ov.copy())
    di = np.argwhere(oi != first[0]); dv = np.argwhere(ov != first[1])
    print('rep', rep, 'T', be.info['threads'], 'differs from rep0: currents', len(di), di[:5].tolist(), [(oi[tuple(x)], first[0][tuple(x)]) for x in di[:3]], 'voltages', len(dv), dv[:5].tolist())
