"""Summarise the rocprofv3 passes written by scripts/pmc_passes.sh into one table (averages per dispatch).
usage: python profiles/summarize_pmc2.py gpurun_out/<dir> > profiles/<round>/pmc_summary.txt"""
import collections, csv, glob, os, sys

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(os.path.join(d, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        name = r["Kernel_Name"]
        name = name.split("::")[-1].split("(")[0] if "anonymous" in name else name.split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
durs = collections.defaultdict(list)
for fn in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        name = r["Kernel_Name"]
        name = name.split("::")[-1].split("(")[0] if "anonymous" in name else name.split("(")[0]
        durs[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("rocprofv3 PMC summary (scripts/pmc_passes.sh: separate passes sq1 / sq2 / sq3 / FETCH_SIZE / WRITE_SIZE / kernel trace); averages per dispatch")
print("SQ_*_CYCLES and SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles; FETCH/WRITE_SIZE in KiB as reported (gfx950: FETCH_SIZE under-reports wide reads 2x)\n")
for k in sorted(acc, key=lambda n: -sum(durs.get(n, [0])) / max(1, len(durs.get(n, [1])))):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    wc = max(1.0, m.get("SQ_WAVE_CYCLES", 1.0))
    us = sum(durs[k]) / len(durs[k]) if durs.get(k) else float("nan")
    f = lambda c: m.get(c, float("nan"))
    print(f"{k}\n   avg {us:9.1f} us  waves={f('SQ_WAVES'):.0f}  VALU={f('SQ_INSTS_VALU'):.3e} SALU={f('SQ_INSTS_SALU'):.3e} LDS={f('SQ_INSTS_LDS'):.3e} "
          f"VMEM_RD={f('SQ_INSTS_VMEM_RD'):.3e} SMEM={f('SQ_INSTS_SMEM'):.3e}\n"
          f"   wave_cycles(quad)={wc:.3e} busy_cycles={f('SQ_BUSY_CYCLES'):.3e}  fractions of wave cycles: VALU_active={f('SQ_ACTIVE_INST_VALU') / wc:.3f} "
          f"SCA_active={f('SQ_ACTIVE_INST_SCA') / wc:.3f} LDS_active={f('SQ_ACTIVE_INST_LDS') / wc:.3f} any_active={f('SQ_ACTIVE_INST_ANY') / wc:.3f} "
          f"wait_any={f('SQ_WAIT_ANY') / wc:.3f} wait_inst_any={f('SQ_WAIT_INST_ANY') / wc:.3f} wait_inst_lds={f('SQ_WAIT_INST_LDS') / wc:.3f}\n"
          f"   LDS_idx_active={f('SQ_LDS_IDX_ACTIVE'):.3e} LDS_bank_conflict={f('SQ_LDS_BANK_CONFLICT'):.3e}  FETCH_KiB={f('FETCH_SIZE'):.0f} WRITE_KiB={f('WRITE_SIZE'):.0f}\n"
          f"   MFMA: insts={f('SQ_INSTS_MFMA'):.3e} mops_bf16={f('SQ_INSTS_VALU_MFMA_MOPS_BF16'):.3e} busy_cycles={f('SQ_VALU_MFMA_BUSY_CYCLES'):.3e} "
          f"GRBM_GUI_ACTIVE={f('GRBM_GUI_ACTIVE'):.3e}")
