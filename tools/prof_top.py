"""Print the top rows of a rocprofv3 kernel_stats.csv (kernel names contain commas): python tools/prof_top.py DIR [N]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))
if not f:
    sys.exit("no kernel_stats.csv under " + sys.argv[1])
rows = list(csv.reader(open(f[0])))[1:]
tot = sum(float(r[2]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("%-78s calls %6s avg %8.1f ns %5.1f%%" % (r[0][:78], r[1], float(r[3]), 100 * float(r[2]) / tot))
