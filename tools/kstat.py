"""print (kernel, calls, average us) of the rows of a rocprofv3 kernel_stats.csv whose name contains argv[2]"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print("   %-48s calls %5s  avg %8.2f us  min %8.2f" % (r["Name"].split("(")[0][:48], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
