"""Per-shard per-kernel times out of a rocprofv3 sqlite trace of tools/shard_perf.py (development aid)."""
import sqlite3, collections, sys
db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = db.execute("select name, start, end from kernels order by start").fetchall()
rank = -1; per = collections.defaultdict(lambda: collections.defaultdict(float))
for name, s, e in rows:
    if "k_synth_fill" in name: rank += 1
    if rank < 0: continue
    k = name.split("(")[0].replace("lsg::", "").replace("void ", "")
    per[k][rank] += (e - s) / 1e6 / steps
n = rank + 1
names = sorted(per, key=lambda k: -sum(per[k].values()))
print("%-24s" % "kernel" + "".join("%7d" % r for r in range(n)))
tot = [0.0] * n
for k in names:
    if "synth" in k: continue
    for r in range(n): tot[r] += per[k][r]
    if sum(per[k].values()) > 0.03 * n: print("%-24s" % k[:24] + "".join("%7.2f" % per[k][r] for r in range(n)))
print("%-24s" % "sum" + "".join("%7.2f" % x for x in tot))
