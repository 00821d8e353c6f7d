"""Per-launch table of the NMS kernels from a rocprofv3 --kernel-trace CSV of tools/prof_nms.py:  python tools/nms_table.py <dir>"""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    nm = r["Kernel_Name"]
    if "nms" in nm or "vote" in nm:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print("%9.1f us  grid %sx%sx%s  %s" % (us, r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"), nm[:64]))
