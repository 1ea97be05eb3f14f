"""Copies a rocprofv3 *_kernel_stats.csv (or counter_collection.csv) into profiles/ with kernel names cut to 110 chars
(rocPRIM template names run to kilobytes).  usage: trim_rocprof.py <in.csv> <out.csv> [kernel substring filter]"""
import csv, sys
src, dst = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else None
rows = list(csv.reader(open(src)))
hdr = rows[0]
ki = hdr.index("Name") if "Name" in hdr else hdr.index("Kernel_Name")
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(hdr)
    for r in rows[1:]:
        if flt and flt not in r[ki]:
            continue
        r[ki] = r[ki][:110]
        w.writerow(r)
