"""Print the kernel timeline of the last boxes from a rocprofv3 --kernel-trace CSV (ms relative to the end)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
window = float(sys.argv[2]) if len(sys.argv) > 2 else 500.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = int(rows[-1]["End_Timestamp"])
for r in rows:
    s = (int(r["Start_Timestamp"]) - last) / 1e6
    e = (int(r["End_Timestamp"]) - last) / 1e6
    if s > -window and e - s > 0.2:
        print(f"{s:9.2f} {e:9.2f} {e - s:8.2f} q{r['Queue_Id']} s{r['Stream_Id']:>2} {r['Kernel_Name'][:28]:28} grid={r['Grid_Size_X']}")
