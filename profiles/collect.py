"""Turn the raw rocprofv3 output of scripts/make_profiles.sh (gpurun_out/<tag>/) into the committed evidence under
profiles/<tag>/:  kernel_stats.csv, kernel_stats_train.csv, pmc_summary.txt, pmc_summary_shell.txt, pmc_summary_mlp.txt,
traffic.json (HBM bytes per kernel launch + which kernels make up one forward frame), mlp_pmc.json, bench.json.
usage: python profiles/collect.py <tag>"""
import collections, csv, glob, json, os, re, shutil, subprocess, sys

tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)


def short(name):
    # (rocprofv3 leaves names with _Float16 / __bf16 parameters mangled: _ZN12_GLOBAL__N_1<len><name>... -- which is how
    # decoder_forward_kernel never matched and round 5's mlp_pmc.json was not written)
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if m:
        n = int(m.group(1))
        base = name[m.end():m.end() + n]
        t = re.match(r"ILb([01])E", name[m.end() + n:])
        return base + (f"<{'true' if t.group(1) == '1' else 'false'}>" if t else "")
    name = name.split("(anonymous namespace)::")[-1] if "anonymous" in name else name
    return name.split("(")[0].strip()


def copy_stats(sub, out):
    fn = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if fn:
        shutil.copy(fn[0], os.path.join(dst, out))


copy_stats("bench", "kernel_stats.csv")
copy_stats("train", "kernel_stats_train.csv")
copy_stats("train_fp32", "kernel_stats_train_fp32.csv")
copy_stats("hd", "kernel_stats_hd.csv")
for f in ("bench.json", "bench_under_rocprof.json", "full_size_errors.txt", "hd_timing.txt", "frame_trace_1M_1024_cube.txt",
          "frame_trace_1M_1024_shell.txt", "backward_blend_counters.txt", "kernel_resources.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
for sub, out in (("pmc", "pmc_summary.txt"), ("pmc_shell", "pmc_summary_shell.txt"), ("pmc_mlp", "pmc_summary_mlp.txt"),
                 ("pmc_hl", "pmc_summary_decoder_fp32.txt")):
    if os.path.isdir(os.path.join(src, sub)):
        txt = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "summarize_pmc2.py"), os.path.join(src, sub)],
                             capture_output=True, text=True).stdout
        open(os.path.join(dst, out), "w").write(txt)


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(os.path.join(src, sub, "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def durations(sub):
    d = collections.defaultdict(list)
    for fn in glob.glob(os.path.join(src, sub, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            d[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return d


# ---- traffic.json: hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE (KiB as reported -> bytes), see the correction note
for sub, workload, out in (("pmc", "1M_1024_cube", "traffic.json"), ("pmc_shell", "1M_1024_shell", "traffic_shell.json")):
    c = counters(sub)
    if not c:
        continue
    dur = durations(sub)
    frames = max([len(v) for k, v in dur.items() if k.startswith("blend_forward")] + [1])   # frames of the traced run
    kern = {}
    for k, m in c.items():
        if "FETCH_SIZE" not in m or "WRITE_SIZE" not in m:
            continue
        kern[k] = {"fetch_KiB_reported": round(m["FETCH_SIZE"]), "write_KiB_reported": round(m["WRITE_SIZE"]),
                   "hbm_bytes": int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024),
                   "valu_wave_insts": m.get("SQ_INSTS_VALU"), "salu_wave_insts": m.get("SQ_INSTS_SALU"),
                   "lds_wave_insts": m.get("SQ_INSTS_LDS"), "avg_us": (sum(dur[k]) / len(dur[k])) if dur.get(k) else None,
                   "launches_per_frame": (len(dur[k]) / frames) if dur.get(k) else None}
    fwd = [k for k in kern if any(k.startswith(p) for p in ("preprocess_kernel", "scan_", "sort_", "rb_", "tilebin_", "duplicate_",
                                                            "ranges_", "blend_forward"))]
    bwd_blend = next((k for k in kern if k.startswith("blend_backward")), "")
    doc = {"workload": workload,
           "source": f"profiles/{tag}/{'pmc_summary.txt' if sub == 'pmc' else 'pmc_summary_shell.txt'} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, "
                     "averages per dispatch; scripts/pmc_passes.sh)",
           "correction": "hbm_bytes = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes): gfx950 FETCH_SIZE reports half of a wide streaming read "
                         "(MI355X_MICROARCH.md, HBM section; calibrated in round 1 on kernels of known traffic: scan_apply reads 3906 KiB "
                         "and reports 1988, preprocess reads 54688 KiB and reports 27359); WRITE_SIZE unscaled",
           "kernels": kern,
           # one STEADY-STATE forward frame: with the two-launch depth sort its eight kernels, once each (the traced run's first
           # frames go through the two-call form and the three-pass sort; their kernels stay in `kernels`)
           "forward_kernels": ({k: 1.0 for k in kern if any(k.startswith(p) for p in (
               "preprocess_kernel<false, true>", "sort_msd_", "rb_level1", "rb_count2", "rb_scan2", "rb_scatter2", "blend_forward"))}
                               if any(k.startswith("sort_msd_finish") for k in kern)
                               else {k: kern[k]["launches_per_frame"] or 1 for k in fwd}),
           "stage_to_kernel": {"blend": next((k for k in kern if k.startswith("blend_forward")), ""), "blend_bwd": bwd_blend,
                               "preprocess": "preprocess_kernel",
                               "duplicate": next((k for k in kern if k.startswith("rb_scatter2")), ""),
                               "sort": next((k for k in kern if k.startswith("sort_msd_finish")),
                                            next((k for k in kern if k.startswith("sort_onesweep")), ""))}}
    json.dump(doc, open(os.path.join(dst, out), "w"), indent=1)

# ---- mlp_pmc.json
c = counters("pmc_mlp")
k = next((n for n in c if n.startswith("decoder_forward_kernel")), None)
if k:
    m = c[k]
    dur = durations("pmc_mlp").get(k, [])
    gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0         # the counter sums the 8 XCDs
    steady = sorted(dur[len(dur) // 2:])     # the second half of the traced run's launches: the first ones run cold (clocks, caches)
    doc = {"kernel": k, "points": 1000000, "kernel_us": steady[len(steady) // 2] if steady else None,
           "kernel_us_all_launches_mean": sum(dur) / len(dur) if dur else None, "launches": len(dur),
           "mfma_insts": m.get("SQ_INSTS_MFMA"), "mfma_mops_bf16": m.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"),
           "mfma_busy_cycles": m.get("SQ_VALU_MFMA_BUSY_CYCLES"), "gui_active_cycles_per_xcd": gui,
           "mfma_busy_frac": (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * gui)) if gui else None,
           "valu_insts": m.get("SQ_INSTS_VALU"),
           "note": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the fraction of SIMD cycles "
                   "with the matrix pipe busy, from the counter pass.  kernel_us = median duration of the second half of the launches in "
                   "the kernel-trace pass of the same script (scripts/mlp_only.py 40): the steady-state kernel, comparable with "
                   "bench.json's decode_render.mlp_ms -- bench.py refuses to replay a profile whose kernel_us is > 20 % off its own timing"}
    json.dump(doc, open(os.path.join(dst, "mlp_pmc.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(dst)))
