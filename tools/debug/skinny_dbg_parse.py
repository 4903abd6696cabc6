import csv, sys, json
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "skinny" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
meta = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
rows = rows[meta["warm"]:]
names = {0: "all", 1: "no mfma", 2: "no decode", 3: "no mfma, no decode", 4: "no row copies", 8: "no split-K exchange", 16: "no epilogue", 31: "stream only"}
for i, (ks, dbg) in enumerate(meta["configs"]):
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0 for r in rows[i * meta["reps"]:(i + 1) * meta["reps"]])
    print(f"ks {ks:2d}  {names[dbg]:24s} median {d[len(d) // 2]:7.2f} us   min {d[0]:7.2f}")
