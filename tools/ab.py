#!/usr/bin/env python3
"""A/B helper (GPU box): run bench.py under several env settings, print the key numbers.
usage: python tools/ab.py "SS_OS_VARIANT=0 SS_XCD_ORDER=0" "SS_OS_VARIANT=3" ..."""
import json, os, subprocess, sys
for spec in sys.argv[1:]:
    env = dict(os.environ)
    for kv in spec.split():
        k, v = kv.split("=", 1)
        env[k] = v
    vals = []
    for rep in range(2):
        out = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "3", "--cpu-positions", "0"],
                             env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(spec, "FAILED", out.stderr[-500:]); break
        j = json.loads(line[-1])
        vals.append((j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["value"], j["roofline"]["xspec_avg_launch_ms"]))
    for v in vals:
        print(f"{spec:45s} ms/step {v[0]:.4f}  k_os avg launch {v[1]*1e3:8.1f} us  frac {v[2]:.4f}  value {v[3]:.0f}  xspec {v[4]*1e3:.1f} us")
