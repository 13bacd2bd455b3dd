#!/usr/bin/env python3
"""Host <-> device rate of the table upload / download entry points (the only ABI calls that take HOST buffers).
The iteration path never uses them: inputs of rs_iterate are device resident.  Numbers go to DESIGN.md section 5."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustsolver_amd as rs  # noqa: E402

t = rs.InfosetTable.create([(3, 1000, 9216, 0, 0)])
R = np.random.default_rng(0).integers(-10**6, 10**6, size=(3, 9216000)).astype(np.int32)
S = np.zeros_like(R)
for name, fn in (("upload_node", lambda: t.upload_node(0, R, S)), ("download_node", lambda: t.download_node(0))):
    fn()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    dt = (time.perf_counter() - t0) / 3
    print("%s: %.1f MB in %.1f ms = %.1f GB/s (pageable host memory)" % (name, 2 * R.nbytes / 1e6, dt * 1e3, 2 * R.nbytes / dt / 1e9))
