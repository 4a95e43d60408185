"""Timeline of the LAST pass of a rocprofv3 --kernel-trace run of bench.py: kernel, queue, start and duration in microseconds relative to the
step's first kernel (lsg::k_seg_static, the start of lsg_load_reads).  usage: python tools/timeline.py <dir holding *_kernel_trace.csv>"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short_kernel_name
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if "k_seg_static" in r[2]]
lo = starts[-1]
t0 = rows[lo][0]
end = 0
for s, e, name, q in rows[lo:]:
    short = short_kernel_name(name).replace("lsg::", "")
    print("%9.1f %9.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short[:70]))
    end = max(end, e)
print("pass: %.1f us" % ((end - t0) / 1e3))
