#!/usr/bin/env python3
"""Decode gpurun_out/trace.bin (OS13_OPT=trace build): per-phase cycle statistics of waves 0 and 4 of workgroup 0."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace.bin", dtype=np.uint32)
for wave in (0, 4):
    rec = raw[wave * 16384:(wave + 1) * 16384].reshape(-1, 2)
    rec = rec[rec[:, 0] != 0]
    t, tag = rec[:, 0].astype(np.int64), rec[:, 1]
    print(f"wave {wave}: {len(rec)} records, span {t.max() - t.min()} ticks")
    # phase durations: from tag a to the next record
    d = np.diff(t)
    for a in sorted(set(tag[:-1])):
        m = tag[:-1] == a
        nxt = tag[1:][m]
        print(f"  tag {a} -> next {np.bincount(nxt).argmax()}: n={m.sum()} mean {d[m].mean():.0f} median {np.median(d[m]):.0f} p90 {np.percentile(d[m], 90):.0f} max {d[m].max()}")
    # first 40 records relative
    print("  first records:", [(int(x - t[0]), int(g)) for x, g in zip(t[:30], tag[:30])])
