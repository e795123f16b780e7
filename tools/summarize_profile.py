"""Summarise a directory of rocprofv3 CSV outputs (kernel trace + PMC passes) per kernel.
Usage: python tools/summarize_profile.py gpurun_out/r03 > profiles/r03_summary.md"""
import csv, glob, os, sys, collections
d = sys.argv[1]
print(f"# rocprofv3 summary of `{d}`\n")
print("Kernel trace: `python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline` (256 frames of 1080p per launch; the driver's own command "
      "with more steps, so that the average is the steady state's).  PMC passes: the same command with `--steps 5 --warmup 2`, one pass "
      "per run; `pmc_many_*`: `python3 tools/prof_run.py 3 4 hybrid many` (BASELINE config 5, 10 000 spheres, 4 frames per launch).\n")
for f in glob.glob(os.path.join(d, "kt", "**", "*kernel_stats.csv"), recursive=True):
    print("## kernel-trace --stats\n\n| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        print(f"| `{r['Name'][:86]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")


# The bench launches the render kernels in two sizes: the timed batches (256 frames) and single frames (work counters, the
# per-launch latency figure, checks).  --stats averages them together; the table below keeps them apart (from the trace itself).
for f in glob.glob(os.path.join(d, "kt", "**", "*kernel_trace.csv"), recursive=True):
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ptr::" not in k: continue
        frames = int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_Z"]))
        groups[(k.split("(")[0][:70], frames if ("render_" in k) else 0)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("\n## kernel trace by launch size (frames per launch = grid z; 0 = not a per-frame grid)\n")
    print("| kernel | frames per launch | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|---|")
    for (k, fr), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        print(f"| `{k}` | {fr} | {len(v)} | {sum(v) / len(v):.0f} | {min(v)} | {max(v)} |")


def short(k):
    k = k.split("(")[0]
    return k[:64]


def big_enough(k, grid):
    # the bench also renders single frames (work counters, per-launch latency): only the batched launches are summarised
    if ("render_fast" in k or "render_ext" in k or "retrace" in k) and "prof_run" not in d:
        return grid >= 10_000_000 or "retrace" in k
    return True


print("\n## PMC counters (mean per dispatch of the batched launches)\n")
print("| pass | kernel | counter | mean per dispatch | dispatches |\n|---|---|---|---|---|")
per = collections.defaultdict(dict)
for p in sorted(glob.glob(os.path.join(d, "pmc_*"))):
    if not os.path.isdir(p): continue
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "ptr::" not in k and "rocprim" not in k and "hipcub" not in k: continue
            if "pmc_many" not in p and not big_enough(k, int(r["Grid_Size"])): continue
            if "retrace" in k and int(r["Grid_Size"]) < 1_000_000 and "many" not in p: continue
            agg[(short(k), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print(f"| {os.path.basename(p)} | `{k}` | {c} | {sum(v)/len(v):.6g} | {len(v)} |")
            per[(os.path.basename(p), k)][c] = sum(v) / len(v)

print("\n## Derived, per kernel\n")
print("Waves per SIMD (time average over the cycles a CU is busy) = SQ_WAVE_CYCLES / SQ_BUSY_CU_CYCLES: calibrated on kernels that hold a known "
      "number of waves on every SIMD (`tools/ubench/occupancy_cal.hip`: 1.000 at 1 wave per SIMD, 2.000 at 2; `SQ_LEVEL_WAVES` and "
      "`SQ_ACCUM_PREV_HIRES` read 0 under rocprofv3 on gfx950, and so do the derived occupancy metrics built on them).\n")
print("| pass | kernel | waves | VALU / wave | SALU / wave | SMEM / wave | waves per SIMD | wait-inst share | LDS instr | LDS bank conflicts |\n|---|---|---|---|---|---|---|---|---|---|")
keys = sorted({k for (_, k) in per})
for k in keys:
    for many in ("", "many_"):
        a = per.get((f"pmc_{many}sq1", k), {}); o = per.get((f"pmc_{many}occ", k), {}); l = per.get(("pmc_sq2", k), {}) if not many else o
        if not a and not o: continue
        w = a.get("SQ_WAVES") or o.get("SQ_WAVES") or 0
        if w < 64: continue
        f = lambda x: f"{x:.1f}" if x is not None else "-"
        occ = o["SQ_WAVE_CYCLES"] / o["SQ_BUSY_CU_CYCLES"] if o.get("SQ_BUSY_CU_CYCLES") else None
        print(f"| {'many' if many else 'bench'} | `{k}` | {w:.0f} | {f(a['SQ_INSTS_VALU'] / w) if a else '-'} | {f(a['SQ_INSTS_SALU'] / w) if a else '-'} | "
              f"{f(a['SQ_INSTS_SMEM'] / w) if a else '-'} | {f'{occ:.2f}' if occ else '-'} | "
              f"{(a['SQ_WAIT_INST_ANY'] / a['SQ_WAVE_CYCLES']):.2f}" if a else "-", end="")
        print(f" | {l.get('SQ_INSTS_LDS', 0):.0f} | {l.get('SQ_LDS_BANK_CONFLICT', 0):.0f} |")
