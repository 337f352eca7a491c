"""Build profiles/rNN_pmc_traffic.json from two rocprofv3 counter runs of the same bench command:
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR_F -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d DIR_W -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python tools/pmc_traffic.py DIR_F DIR_W OUT.json
Per kernel: KB per dispatch as reported (FETCH_SIZE / WRITE_SIZE are in KB).  legendre_span_bytes = HBM bytes one
9-map Legendre span (all synthesis launches of one matvec, or all adjoint launches) moves; spans are counted by the
k_band_prep dispatches (one per matvec and plan)."""
import collections, csv, glob, json, re, sys

def load(d, name):
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            acc[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return acc, cnt

fa, fc = load(sys.argv[1], "FETCH_SIZE")
wa, wc = load(sys.argv[2], "WRITE_SIZE")
kern = {}
for k in sorted(set(fa) | set(wa)):
    if not k.startswith("cmdr::"):
        continue
    kern[k] = {"dispatches": int(max(fc[k], wc[k])),
               "FETCH_SIZE_KB_per_dispatch": fa[k] / max(fc[k], 1), "WRITE_SIZE_KB_per_dispatch": wa[k] / max(wc[k], 1)}
nspan_mv = max(fc.get("cmdr::k_band_prep", 0), 1)                     # matvecs
def span(prefix, nspan):
    return sum((fa[k] + wa[k]) * 1024.0 for k in kern if k.startswith(prefix)) / nspan
nspan_adj = nspan_mv + max(fc.get("cmdr::k_pix", 0) // 9, 0)          # the RHS adds one adjoint span per 9 k_pix launches
out = {"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py "
                  "--steps 1 --warmup 0 --no-cpu-baseline",
       "note": "FETCH_SIZE/WRITE_SIZE in KB per dispatch as reported.  MI355X_MICROARCH.md (HBM): FETCH_SIZE halves 16-B/lane "
               "coalesced streaming reads, other widths are uncalibrated; the Legendre kernels read through 8-B/lane vector "
               "loads and scalar-cache requests, so no x2 correction is applied and the absolute numbers carry that caveat",
       "kernels": kern,
       "legendre_span_bytes": {"synth_9maps": span("cmdr::k_leg_synth", nspan_mv),
                               "adjoint_9maps": span("cmdr::k_leg_adj<4", nspan_adj) + span("cmdr::k_leg_adj_mx", nspan_adj)}}
ls = out["legendre_span_bytes"]
ls["mean"] = 0.5 * (ls["synth_9maps"] + ls["adjoint_9maps"])
mx = next((v for k, v in kern.items() if k.startswith("cmdr::k_leg_adj_mx")), None)   # template arguments vary
if mx:   # the dominant kernel of round 2: one launch = the Legendre adjoint of 8 maps on the matrix unit
    out["adjoint_launch_bytes"] = {"kernel": "cmdr::k_leg_adj_mx",
                                   "mean": (mx["FETCH_SIZE_KB_per_dispatch"] + mx["WRITE_SIZE_KB_per_dispatch"]) * 1024.0,
                                   "fetch": mx["FETCH_SIZE_KB_per_dispatch"] * 1024.0, "write": mx["WRITE_SIZE_KB_per_dispatch"] * 1024.0}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(ls))
