"""Timeline of ONE forward frame from a rocprofv3 kernel trace (begin / end of every kernel, gap to its predecessor):
    rocprofv3 --kernel-trace -d <dir> -o p --output-format csv -- python scripts/fwd_only.py 1M_1024_cube 12
    python scripts/frame_trace.py <dir>/p_kernel_trace.csv [frame index from the end, default 3]
Prints the kernels of that frame in launch order: start (us from the frame's first kernel), duration, gap since the end of
the previous kernel, grid (workgroups), and the sums -- what a frame's latency is made of."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
starts = [i for i, r in enumerate(rows) if name(r).startswith("preprocess_kernel")]
i0, i1 = starts[-back - 1], starts[-back]
fr = rows[i0:i1]
t0 = int(fr[0]["Start_Timestamp"])
prev_end, tot_dur, tot_gap = None, 0.0, 0.0
print(f"{'kernel':58s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>7s} {'workgroups':>10s}")
for r in fr:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    wg = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) // max(1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1))
    print(f"{name(r)[:58]:58s} {(s - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f} {gap:7.2f} {wg:10d}")
    tot_dur += (e - s) / 1e3; tot_gap += gap; prev_end = e
print(f"{len(fr)} kernels: durations {tot_dur:.1f} us + gaps {tot_gap:.1f} us = {(prev_end - t0) / 1e3:.1f} us from the first kernel's start to the last one's end")
nxt = int(rows[i1]["Start_Timestamp"])
print(f"next frame's first kernel starts {(nxt - prev_end) / 1e3:.2f} us after this frame's last kernel ended")
