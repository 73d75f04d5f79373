"""Development aid: print the kernel timeline of one scan from a rocprofv3 rocpd database."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = list(db.execute("select name,start,end,queue_id,stream_id from kernels order by start"))


def short(n):
    m = re.search(r"(\w+)\(", n)
    return m.group(1) if m else n[:40]


idx = [i for i, r in enumerate(rows) if "pre_transform" in r[0]]
i0, i1 = idx[which], idx[which + 1]
t0 = rows[i0][1]
prev = {}
busy = {}
for r in rows[i0:i1]:
    p = prev.get(r[4], t0)
    print(f"{(r[1]-t0)/1e3:9.2f} dur={(r[2]-r[1])/1e3:7.2f} gap={(r[1]-p)/1e3:7.2f} s={r[4]} {short(r[0])}")
    prev[r[4]] = r[2]
    busy[r[4]] = busy.get(r[4], 0) + (r[2] - r[1])
print("scan period us", (rows[i1][1] - rows[i0][1]) / 1e3, "busy us per stream", {k: v / 1e3 for k, v in busy.items()})
per = [(rows[idx[k + 1]][1] - rows[idx[k]][1]) / 1e3 for k in range(10, len(idx) - 1)]
print("mean period", sum(per) / len(per))
