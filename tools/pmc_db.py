#!/usr/bin/env python3
"""Print per-dispatch duration and PMC counters from a rocprofv3 results .db (rocpd sqlite)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
rows = list(cur.execute("select * from counters_collection"))
import collections
by = collections.OrderedDict()
for r in rows:
    d = dict(zip(cols, r))
    key = (d.get("dispatch_id"), d.get("kernel_name") or d.get("name"))
    e = by.setdefault(key, {"dur_us": (d.get("end", 0) - d.get("start", 0)) / 1e3})
    e[d.get("counter_name")] = e.get(d.get("counter_name"), 0) + (d.get("value") or 0)
for (did, name), e in by.items():
    print(did, (name or "")[:60], {k: (round(v, 1) if isinstance(v, float) else v) for k, v in e.items()})
