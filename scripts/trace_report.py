#!/usr/bin/env python3
"""Summarise gpurun_out/trace_<layer>.txt (QCNN_TRACE experiments): median cycles of the builder wave's and
the gather wave's phases inside one workgroup."""
import sys
import numpy as np

for path in sys.argv[1:]:
    rows = [list(map(int, l.split())) for l in open(path)]
    a = np.zeros(1 << 16, dtype=np.int64)
    for r in rows:
        a[r[0]:r[0] + 8] = r[1:]
    b = a[:32768].reshape(-1, 8); b = b[b[:, 0] > 0]
    g = a[32768:].reshape(-1, 4); g = g[g[:, 0] > 0]
    d = np.diff(b[:, :6], axis=1); per = np.diff(b[:, 0])
    print(path, "builder pairs", len(b), "gather stages", len(g))
    print("  builder median cycles: storeA %d loadA %d barrierA %d storeB %d loadB %d | pair period %d" %
          (tuple(np.median(d, axis=0)) + (np.median(per),)))
    dg = np.diff(g, axis=1); pg = np.diff(g[:, 0])
    print("  gather  median cycles: bcast+prefetch %d gather %d barrier-wait %d | stage period %d" %
          (tuple(np.median(dg, axis=0)) + (np.median(pg),)))
    print("  gather rows 4..9:", dg[4:10].tolist())
