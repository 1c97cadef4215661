"""Print the kernel timeline (durations and gaps, us) of the last solves in a rocprofv3 kernel-trace CSV."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -40:]
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("clc::", "")[:28]
    print("%-28s dur %7.2f  gap %7.2f" % (name, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
