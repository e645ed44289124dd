"""Prints queue id, grid size and kernel name of every dispatch of a rocprofv3 kernel trace, in start order."""
import csv
import glob
import sys

for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    print(path, len(rows), "dispatches; columns:", list(rows[0].keys()) if rows else None)
    for r in rows:
        print(r.get("Queue_Id"), r.get("Stream_Id", ""), r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", ""),
              r["Kernel_Name"][:40])
