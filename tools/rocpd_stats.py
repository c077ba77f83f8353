"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) as a per-kernel stats table.
usage: python tools/rocpd_stats.py <results.db> [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
tot = sum(r[2] for r in rows)
lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
for r in rows:
    n = r[0].split("(")[0][:90]
    lines.append(f"\"{n}\",{r[1]},{r[2]},{r[3]:.1f},{r[4]},{r[5]},{100.0 * r[2] / tot:.2f}")
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
print(out)
