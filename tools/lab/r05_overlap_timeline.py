"""Post-process a `rocprofv3 --kernel-trace` csv of `bench.py --no-secondary` (three render streams): print the timeline of a dozen consecutive
renders -- per render its stream (queue), when its spectra kernel and its persistent kernel started / ended relative to the previous persistent
kernel's end -- and the period between persistent-kernel ENDS (= the step).  usage: python tools/lab/r05_overlap_timeline.py <kernel_trace.csv>"""
import csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r["Queue_Id"]) for r in rows
            if r["Kernel_Name"].startswith(("k_os13_asm", "k_xspec13")))
osl = [e for e in ev if e[2] == "k_os13_asm"]
xsl = [e for e in ev if e[2] == "k_xspec13"]
# the densest stretch: 200 consecutive persistent launches (by END time) with the smallest span = inside the sustained, overlapped windows
osl.sort(key=lambda e: e[1])
best = min(range(0, max(1, len(osl) - 200)), key=lambda i: osl[i + 199][1] - osl[i][1]) if len(osl) > 220 else 0
seg = osl[best:best + 200]
span = seg[-1][1] - seg[0][1]
dur = [e[1] - e[0] for e in seg]
print(f"{len(osl)} persistent launches in the trace; densest 200 by end time: {span / 199 / 1e3:.2f} us per render (span / 199); "
      f"a launch lasts {statistics.median(dur) / 1e3:.1f} us from its first workgroup in to its last workgroup out (median) -- the kernel ALONE takes ~163: "
      f"two to three launches share the compute units at any moment")
print("queues (streams) used:", sorted({e[3] for e in seg}))
print("render | queue | spectra start .. end | persistent start .. end    (us, relative to the end of render 100's persistent launch)")
t0 = seg[100][1]
for k in range(100, 112):
    s, e, _, q = seg[k]
    xs = [x for x in xsl if x[3] == q and x[1] <= s]
    x = xs[-1] if xs else None
    xs_txt = f"{(x[0] - t0) / 1e3:9.2f} ..{(x[1] - t0) / 1e3:9.2f}" if x else "        ?"
    print(f"{k:6d} | {q:>5s} | {xs_txt} | {(s - t0) / 1e3:9.2f} ..{(e - t0) / 1e3:9.2f}")
