#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/collect_profiles.sh into the committed summaries:

  profiles/<round>_rocprofv3_kernel_stats.csv   per-kernel Calls / AverageNs (kernel-trace --stats)
  profiles/<round>_pmc_hbm.json                 FETCH_SIZE / WRITE_SIZE per launch, per kernel
  profiles/roofline_traffic.json                label -> HBM bytes per launch (bench.py reads it)

Corrections (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE and WRITE_SIZE are reported in KiB-like
units of 1 KB by rocprofv3's derived metric; on gfx950 FETCH_SIZE under-reports wide coalesced
reads by exactly 2x, so it is doubled.  WRITE_SIZE is uncalibrated and used as reported.
"""
import csv, glob, json, os, sys, collections

LABELS = [("ntt_kernel<true", "ntt_inv"), ("ntt_kernel<false", "ntt_fwd"), ("ntt_global_kernel", "ntt_global"),
          ("tensor_intt_kernel", "tensor_intt"), ("ks_fused_kernel", "key_switch_fused"),
          ("scale_kernel<4", "scale_extend"), ("scale_kernel<9", "scale_down"),  # C2: L=4 -> K=9 and back (<NF, PLAIN> since round 3)
          ("scale_kernel", "scale"), ("copy_rows_kernel", "copy_rows"), ("tensor_kernel", "tensor"),
          ("switch_down_kernel", "switch_down"), ("substitute_kernel", "substitute"),
          ("dot_kernel", "dot_product"), ("synth_kernel", "synth")]


def label(name):
    for pat, lab in LABELS:
        if pat in name:
            return lab
    return None


def pmc(dirname, counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            lab = label(r["Kernel_Name"])
            if lab:
                tot[lab][0] += 1
                tot[lab][1] += float(r["Counter_Value"])
    return tot


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(prof, f"{rnd}_rocprofv3_kernel_stats.csv"), "w") as o:
            w = csv.writer(o)
            w.writerow(["Label", "Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                nm = r["Name"]
                w.writerow([label(nm) or "", nm.split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                            r["Percentage"], r["MinNs"], r["MaxNs"]])
    fetch = pmc(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
    write = pmc(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    out, traffic = {}, {}
    for lab in sorted(set(fetch) | set(write)):
        fl, fv = fetch.get(lab, [0, 0.0])
        wl, wv = write.get(lab, [0, 0.0])
        # rocprofv3 reports both in KB (1024 B); FETCH_SIZE doubled per the guide's gfx950 correction
        fb = (fv / fl * 1024 * 2) if fl else None
        wb = (wv / wl * 1024) if wl else None
        out[lab] = dict(launches_fetch=fl, launches_write=wl, fetch_bytes_per_launch_corrected=fb,
                        write_bytes_per_launch=wb, fetch_raw_kb_per_launch=(fv / fl if fl else None))
        if fb is not None and wb is not None:
            traffic[lab] = int(fb + wb)
    json.dump(out, open(os.path.join(prof, f"{rnd}_pmc_hbm.json"), "w"), indent=1)
    traffic["_source"] = (f"profiles/{rnd}_pmc_hbm.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, "
                          "separate passes over `bench.py --steps K --warmup 5 --no-cpu --no-extras` (tools/collect_profiles.sh), bytes per launch")
    json.dump(traffic, open(os.path.join(prof, "roofline_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
