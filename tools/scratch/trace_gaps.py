import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last forward: find last LSTM kernel, walk back to previous LSTM kernel
idx = [i for i, r in enumerate(rows) if "lstm2_" in r["Kernel_Name"] and "seq" not in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
seg = rows[a + 1:b + 1]
t0 = int(rows[a]["End_Timestamp"])
busy = 0
prev_end = t0
gaps = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    gaps.append((s - prev_end, r["Kernel_Name"][:50]))
    prev_end = max(prev_end, e)
total = prev_end - t0
print("kernels", len(seg), "span us %.1f busy us %.1f  idle us %.1f" % (total / 1e3, busy / 1e3, (total - busy) / 1e3))
big = sorted(gaps, reverse=True)[:6]
for g, n in big: print("  gap %.1f us before %s" % (g / 1e3, n))
