"""Summarise a directory of rocprofv3 CSV outputs (kernel trace + PMC passes) per kernel.
Usage: python tools/summarize_profile.py gpurun_out/r01 > profiles/r01_summary.md"""
import csv, glob, os, sys, collections
d = sys.argv[1]
print(f"# rocprofv3 summary of `{d}`\n")
print("Command profiled: `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline` (256 frames 1080p per launch)\n")
for f in glob.glob(os.path.join(d, "kt", "**", "*kernel_stats.csv"), recursive=True):
    print("## kernel-trace --stats\n\n| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        print(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")
print("\n## PMC counters (mean per dispatch, dispatches of the timed/warm steps)\n")
print("| pass | kernel | counter | mean per dispatch | dispatches |\n|---|---|---|---|---|")
for p in sorted(glob.glob(os.path.join(d, "pmc_*"))):
    if not os.path.isdir(p): continue
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "ptr::" not in k: continue
            if ("render_fast" in k or "render_ext" in k) and int(r["Grid_Size"]) < 10_000_000: continue   # skip 1-frame calls
            agg[(k.split("(")[0][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print(f"| {os.path.basename(p)} | `{k}` | {c} | {sum(v)/len(v):.6g} | {len(v)} |")
