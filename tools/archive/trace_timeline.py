"""Merged kernel + memory-copy timeline (us, relative) of the last launches in a rocprofv3 output directory."""
import csv, glob, sys
d, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("clc::", "")[:30]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:24]))
ev.sort()
ev = ev[-last:]
t0, prev = ev[0][0], None
for s, e, name in ev:
    print("%9.2f  %-30s dur %7.2f  gap %7.2f" % ((s - t0) / 1e3, name, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
