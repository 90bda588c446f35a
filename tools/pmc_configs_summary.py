#!/usr/bin/env python3
"""Per-kernel HBM traffic and duration of the C3 / C5 workloads from the outputs of tools/collect_configs_pmc.sh.
FETCH_SIZE is doubled (gfx950 correction of MI355X_MICROARCH.md: wide coalesced reads are tallied at half their
bytes); both counters are in KiB.  Prints JSON: {cfg: {kernel: {launches, fetch_bytes_per_launch,
write_bytes_per_launch, avg_ns}}}."""
import collections, csv, glob, json, os, sys


def short(name):
    n = name.split("(")[0]
    return n.replace("void fhe::k::", "").replace("fhe::k::", "")


def pmc(d, counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "fhe" in r["Kernel_Name"]:
                e = tot[short(r["Kernel_Name"])]
                e[0] += 1
                e[1] += float(r["Counter_Value"])
    return tot


def main():
    src = sys.argv[1]
    out = {}
    for cfg in ("c3", "c5"):
        fetch, write = pmc(os.path.join(src, cfg + "_fetch"), "FETCH_SIZE"), pmc(os.path.join(src, cfg + "_write"), "WRITE_SIZE")
        dur = {}
        for f in glob.glob(os.path.join(src, cfg + "_stats", "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "fhe" in r["Name"]:
                    dur[short(r["Name"])] = dict(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]))
        res = {}
        for k in sorted(set(fetch) | set(write)):
            fl, fv = fetch.get(k, [0, 0.0])
            wl, wv = write.get(k, [0, 0.0])
            res[k] = dict(launches=fl or wl,
                          fetch_bytes_per_launch=int(fv / fl * 1024 * 2) if fl else None,
                          write_bytes_per_launch=int(wv / wl * 1024) if wl else None,
                          avg_ns=dur.get(k, {}).get("avg_ns"))
        out[cfg] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
