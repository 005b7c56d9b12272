"""Per-kernel average of the PMC counters in a rocprofv3 results .db."""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if "_0000" not in t and not t[-3:].isdigit()])
for t in ("counters_collection", "pmc_events", "rocpd_pmc_event"):
    if t in tabs:
        cols = [d[1] for d in con.execute("pragma table_info(%s)" % t)]
        print(t, cols)
        for r in con.execute("select * from %s limit 2" % t):
            print("  ", r)
if "counters_collection" in tabs:
    cols = [d[1] for d in con.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    d = collections.defaultdict(list)
    for name, cname, val in con.execute("select %s, counter_name, value from counters_collection" % kcol):
        d[(name[:90], cname)].append(val)
    for (name, cname), v in sorted(d.items()):
        print("%-90s %-14s n=%5d avg=%14.1f" % (name, cname, len(v), sum(v) / len(v)))
