import csv, sys, glob, collections
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k, d in acc.items():
    if "wino" not in k and "gemm" not in k: continue
    print(k, {c: "%.3e" % v for c, v in d.items()})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        print("   mfma busy per SIMD / elapsed = %.3f" % ((d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (d["GRBM_GUI_ACTIVE"] / 8)))
    if "SQ_WAVE_CYCLES" in d:
        w = d["SQ_WAVE_CYCLES"]
        print("   of wave cycles:", {c: "%.3f" % (d[c] / w) for c in d if c.startswith("SQ_") and c != "SQ_WAVE_CYCLES"})
