import csv, glob, sys
path = glob.glob(sys.argv[1] + "/*kernel_stats.csv")[0]
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU kernel ms per iter: %.1f   kernels per iter: %.0f" % (tot / 1e6 / div, sum(int(r["Calls"]) for r in rows) / div))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-86s %7.1f/it %8.2f ms/it %5.1f%% avg %8.1f us" % (r["Name"][:86], int(r["Calls"]) / div, float(r["TotalDurationNs"]) / 1e6 / div, float(r["Percentage"]), float(r["AverageNs"]) / 1e3))
