#!/usr/bin/env python3
"""Per-launch HBM traffic of the conv kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE),
following /opt/skills/guides/MI355X_MICROARCH.md "HBM": separate passes, values in KB, FETCH_SIZE doubled on
gfx950 for 16-byte-per-lane reads (the 128-byte requests are tallied at 64 bytes).

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/pmc_traffic.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(d, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[k] += float(row["Counter_Value"])
            n[k] += 1
    return tot, n


def main():
    fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
    write, nw = collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(fetch, key=lambda k: -fetch[k]):
        if nf[k] == 0 or k not in write or not k.startswith("k_"):       # the library's own kernels
            continue
        rd = 2.0 * fetch[k] * 1024.0 / nf[k]          # KB -> bytes, x2 (gfx950 wide-read correction)
        wr = write[k] * 1024.0 / max(nw[k], 1)
        out[k] = {"launches_profiled": nf[k], "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                  "hbm_bytes_per_launch": round(rd + wr)}
    tag = sys.argv[3] if len(sys.argv) > 3 else None
    print(json.dumps({"capture": ("profiles/%s_pmc_traffic.json" % tag) if tag else "untagged capture", "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 per the gfx950 "
                                "note of MI355X_MICROARCH.md; command: " + os.environ.get("PMC_FLAGS_NOTE", "bench.py (flags not recorded)"),
                      "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
