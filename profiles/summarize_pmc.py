"""Summarise rocprofv3 --pmc counter_collection CSVs (one per pass) into one text table.
usage: python profiles/summarize_pmc.py <sq.csv> <fetch.csv> <write.csv> "<title>" > profiles/<round>/pmc_summary.txt"""
import collections
import csv
import sys


def load(fn):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fn)):
        name = r["Kernel_Name"]
        if "anonymous" not in name:
            continue
        name = name.split("::")[1].split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


sq, fe, wr = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
print(sys.argv[4] if len(sys.argv) > 4 else "rocprofv3 PMC summary")
print("separate passes: [SQ_* x8] [FETCH_SIZE] [WRITE_SIZE]; averages per dispatch; FETCH/WRITE_SIZE in KiB as reported "
      "(gfx950: FETCH_SIZE under-reports wide streaming reads by 2x, MI355X_MICROARCH.md HBM section)\n")
for k, d in sq.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    avg = lambda a, key: sum(a[k][key]) / max(1, len(a[k][key]))
    print(f"{k:28s} waves={m['SQ_WAVES']:.0f} VALU_insts={m['SQ_INSTS_VALU']:.3e} LDS_insts={m['SQ_INSTS_LDS']:.3e} "
          f"wave_cycles(quad)={m['SQ_WAVE_CYCLES']:.3e} VALU_active/wave_cyc={m['SQ_ACTIVE_INST_VALU'] / max(1, m['SQ_WAVE_CYCLES']):.3f} "
          f"wait_inst_any/wave_cyc={m['SQ_WAIT_INST_ANY'] / max(1, m['SQ_WAVE_CYCLES']):.3f} "
          f"FETCH_KiB={avg(fe, 'FETCH_SIZE'):.0f} WRITE_KiB={avg(wr, 'WRITE_SIZE'):.0f}")
