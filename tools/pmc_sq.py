"""Condense rocprofv3 --pmc SQ_* counter runs (one directory per pass) into profiles/rNN_pmc_sq.json: per kernel, the
per-dispatch mean of every counter collected.  Usage: python tools/pmc_sq.py OUT.json DIR [DIR ...]"""
import collections, csv, glob, json, re, sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            if not k.startswith("cmdr::k_leg") and not k.startswith("cmdr::k_ring<2>"):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
out = {"command": "rocprofv3 --pmc <SQ counters, four per pass> --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
                  "--no-cpu-baseline --no-extras   (tools/collect_profiles.sh)",
       "note": "per-dispatch means; SQ_* counters are summed over the shader engines",
       "kernels": {k: {c: v / cnt[k][c] for c, v in sorted(cs.items())} | {"dispatches": max(cnt[k].values())}
                   for k, cs in sorted(acc.items())}}
for k, cs in out["kernels"].items():
    if "SQ_INSTS_VALU" in cs and "SQ_BUSY_CYCLES" in cs and cs["SQ_BUSY_CYCLES"]:
        pass
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, cs in out["kernels"].items():
    print(k, {c: "%.3g" % v for c, v in cs.items()})
