import csv, glob, sys
tr = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(tr)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last pass: find last occurrence of im2col kernel
idx = [i for i, r in enumerate(rows) if "rgb_s2" in r["Kernel_Name"] or "im2col" in r["Kernel_Name"]]
start = idx[-1]
tot = 0
for r in rows[start:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    n = r["Kernel_Name"].replace("lla::(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{d:9.1f} us  grid {r['Grid_Size_X']:>9}  {n}")
print("total", tot)
