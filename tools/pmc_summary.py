#!/usr/bin/env python3
"""HBM bytes per launch of the large kernels from two rocprofv3 counter passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`,
each in its own run, --kernel-trace only: gpurun refuses counters together with trace domains) of `bench.py --log-domain L`:
    python tools/pmc_summary.py <fetch dir> <write dir> <out.json> [L]
Writes the summary bench.py reads for `roofline.traffic` (--pmc-summary, or the newest profiles/r*_pmc_summary.json).
Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes (HBM section): the counters count in KB; on gfx950 FETCH_SIZE
tallies a 128-byte request at 64 bytes, so kernels whose reads are wide coalesced streams or whole 128-byte lines are doubled
(DOUBLED below); kernels that gather 64-byte records (half lines) are taken as counted -- which under-counts their coalesced
key / value reads. WRITE_SIZE is taken as counted."""
import collections
import csv
import glob
import json
import sys

DOUBLED = ("ntt_pass_kernel", "segment_accumulate_kernel<G2Cfg>", "radix_pass_kernel", "transpose_entries_kernel", "radix_hist_kernel",
           "bucket_bounds_kernel")


def per_kernel(d):
    fs = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
    if not fs:
        raise RuntimeError("no counter_collection.csv under " + d)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("ug::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return {k: v / n for k, (n, v) in agg.items()}


def summarize(fetch_dir, write_dir, log, how="separate passes, tools/run_r4_prof.sh"):
    fetch, write = per_kernel(fetch_dir), per_kernel(write_dir)
    log = str(log)
    out = {"_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (%s) on `bench.py --log-domain %s --mix U "
                       "--host-threads 1`, per launch, in bytes (counter unit KB x 1024); gfx950 correction: FETCH_SIZE doubled for kernels "
                       "that read coalesced streams or whole 128-byte lines (fetch_doubled), as counted for 64-byte gathers "
                       "(tools/pmc_summary.py)" % (how, log), log: {}}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, 0) + write.get(k, 0))):
        if fetch.get(k, 0) + write.get(k, 0) < 1024:          # below 1 MB per launch
            continue
        name = k.replace("<false, 4>", "<false>").replace("<true, 4>", "<true>")
        name = name.replace("ntt_pass_kernel<true>", "ntt_pass_kernel").replace("ntt_pass_kernel<false>", "ntt_pass_kernel")    # (Shoup / Montgomery form)
        dbl = any(name.startswith(p) or p in name for p in DOUBLED)
        out[log][name] = {"fetch": int(fetch.get(k, 0) * 1024 * (2 if dbl else 1)), "write": int(write.get(k, 0) * 1024),
                          "fetch_raw_kb": fetch.get(k, 0), "write_raw_kb": write.get(k, 0), **({"fetch_doubled": True} if dbl else {})}
    return out


if __name__ == "__main__":
    log = sys.argv[4] if len(sys.argv) > 4 else "24"
    out = summarize(sys.argv[1], sys.argv[2], log)
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in list(out[log].items())[:14]:
        print("%-60s fetch %8.3f GB  write %8.3f GB%s" % (k[:60], v["fetch"] / 1e9, v["write"] / 1e9, "  (fetch doubled)" if v.get("fetch_doubled") else ""))
