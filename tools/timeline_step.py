"""Print the GPU timeline (kernels + copies) of ONE batched env step from a rocprofv3 trace directory:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 tools/time_vecenv.py 128 0 1
   python tools/timeline_step.py DIR [marker-kernel-substring [occurrence]]"""
import csv, glob, sys
d = sys.argv[1]
mark = sys.argv[2] if len(sys.argv) > 2 else "smooth_kernel"
ev = []
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?") + " " + r["Kernel_Name"].split("(")[0][-60:]))
for f in glob.glob(f"{d}/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Size", r.get("Bytes", ""))))
ev.sort()
marks = [i for i, e in enumerate(ev) if mark in e[2]]
at = int(sys.argv[3]) if len(sys.argv) > 3 else -3     # which occurrence of the marker starts the window
a, b = marks[at], marks[at + 1]
t0 = ev[a][0]
busy = 0
prev_end = t0
print(f"step window {(ev[b][0] - t0) / 1e3:.1f} us, {b - a} events")
for s, e, n in ev[a:b]:
    gap = (s - prev_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:7.1f}  {n}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"busy {busy / 1e3:.1f} us of {(ev[b][0] - t0) / 1e3:.1f}")
