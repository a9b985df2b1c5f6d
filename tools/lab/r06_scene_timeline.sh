#!/bin/bash
# per-scene GPU timeline of config 4 at N = 1 with and without the gather bookkeeping (kernel trace): period, gaps, K1 placement
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r06aw}; mkdir -p $OUT
for a in gather nogather; do
  extra=""; [ $a = nogather ] && extra="--no-gather"
  timeout 600 rocprofv3 --kernel-trace -d $OUT/tr_$a -o t -f csv -- python bench.py --config cfg4 --steps 48 $extra --cpu-seconds 0 > $OUT/run_$a.log 2>&1
  f=$(find $OUT/tr_$a -name "*kernel_trace.csv" | head -1)
  python - "$f" $a <<'PY'
import csv, sys, statistics as st
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Stream_Id") or r.get("Queue_Id")))
rows.sort()
os13 = [i for i, r in enumerate(rows) if r[2].startswith("k_os13_asm")]
per = [(rows[b][0] - rows[a][0]) / 1e3 for a, b in zip(os13[:-1], os13[1:])]
# the longest run of near-constant periods = the timed scenes
main = [p for p in per if 700 < p < 1500]
print(sys.argv[2], "scenes", len(main), "period median %.1f us  mean %.1f  p10 %.1f p90 %.1f" % (st.median(main), st.mean(main), sorted(main)[len(main)//10], sorted(main)[9*len(main)//10]))
def dur(prefix):
    d = [(r[1] - r[0]) / 1e3 for r in rows if r[2].startswith(prefix)]
    return "%s n=%d median %.1f" % (prefix, len(d), st.median(d)) if d else prefix + " none"
print("   ", "; ".join(dur(p) for p in ("k_rir_synth_batch", "k_xspec13_multi", "k_os13_asm", "k_kw_fused32", "k_scale_sums", "k_mix_onepass4", "k_gate", "k_lufs_result", "k_block_power")))
# gap between the end of a scene's mix and the start of the next scene's spectra launch; and from the end of the spectra launch to the render kernel
gaps, g2, g3 = [], [], []
for i, r in enumerate(rows):
    if r[2].startswith("k_mix_onepass4"):
        nx = next((q for q in rows[i + 1:] if q[2].startswith("k_xspec13_multi")), None)
        if nx: gaps.append((nx[0] - r[1]) / 1e3)
    if r[2].startswith("k_xspec13_multi"):
        nx = next((q for q in rows[i + 1:] if q[2].startswith("k_os13_asm")), None)
        if nx: g2.append((nx[0] - r[1]) / 1e3)
    if r[2].startswith("k_os13_asm"):
        nx = next((q for q in rows[i + 1:] if q[2].startswith("k_kw_fused32")), None)
        if nx: g3.append((nx[0] - r[1]) / 1e3)
f = lambda v: "median %.1f p90 %.1f" % (st.median(v), sorted(v)[9 * len(v) // 10]) if v else "-"
print("    gap mix end -> next spectra start:", f(gaps), "| spectra end -> render start:", f(g2), "| render end -> K-weighting start:", f(g3))
names = {}
for r in rows: names[r[2]] = names.get(r[2], 0) + 1
print("    kernels:", {k: v for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:16]})
PY
  rm -rf $OUT/tr_$a
done
