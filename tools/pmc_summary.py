"""Mean per-dispatch PMC values from a rocprofv3 rocpd db. usage: python tools/pmc_summary.py db [kernel substring]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(list)
for k, name, v in c.execute("select kernel_name, counter_name, value from counters_collection"):
    if sub in k:
        acc[(k[:60], name)].append(v)
for (k, name), vs in sorted(acc.items()):
    print(f"{k:60s} {name:28s} n={len(vs):3d} mean={sum(vs)/len(vs):16.1f}")
