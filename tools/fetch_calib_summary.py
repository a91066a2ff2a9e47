"""Turns the two rocprofv3 --pmc passes over tools/fetch_calib (FETCH_SIZE, then WRITE_SIZE; tools/pmc_pass.sh) into calibration factors:
bytes really moved / bytes the counter reports (FETCH_SIZE / WRITE_SIZE are in KiB), per access shape.
usage: python tools/fetch_calib_summary.py <pass-1 dir> <pass-2 dir> --out profiles/r05_fetch_calibration.json"""
import collections
import csv
import glob
import json
import os
import sys

BYTES = 327680 * 768 * 2


def per_kernel(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    args = sys.argv[1:]
    out = args[args.index("--out") + 1]
    dirs = [a for i, a in enumerate(args) if not a.startswith("--") and (i == 0 or args[i - 1] != "--out")]
    merged = collections.defaultdict(dict)
    for d in dirs:
        for k, cs in per_kernel(d).items():
            merged[k].update(cs)
    res = {"what": "bytes moved / bytes reported, per access shape, on exactly %d bytes per launch (3 launches each, mean); FETCH_SIZE / WRITE_SIZE in KiB" % BYTES,
           "bytes_per_launch": BYTES, "kernels": {}}
    for k, cs in sorted(merged.items()):
        if not k.startswith("calib_"):
            continue
        e = {"counters": {c: round(v, 1) for c, v in cs.items()}}
        if k.startswith("calib_read") and cs.get("FETCH_SIZE"):
            e["read_factor"] = round(BYTES / (cs["FETCH_SIZE"] * 1024.0), 4)
        if k.startswith("calib_write") and cs.get("WRITE_SIZE"):
            e["write_factor"] = round(BYTES / (cs["WRITE_SIZE"] * 1024.0), 4)
        res["kernels"][k] = e
    g = lambda k, f: res["kernels"].get(k, {}).get(f)
    res["factors"] = {"read_stream": g("calib_read_stream", "read_factor"), "read_rowseg": g("calib_read_rowseg", "read_factor"),
                      "read_lds_dma": g("calib_read_lds_dma", "read_factor"), "write_stream": g("calib_write_stream", "write_factor"),
                      "write_rowseg": g("calib_write_rowseg", "write_factor")}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res["factors"]))


if __name__ == "__main__":
    main()
