"""profiles/<tag>/frame_rooflines.txt: every kernel of ONE traced forward frame against the HBM roofline and the VALU-issue
roofline (durations: frame_trace_1M_1024_cube.txt; HBM bytes and VALU wave-instructions per launch: traffic.json).
usage: python profiles/frame_rooflines.py <tag>"""
import json, os, re, sys
tag = sys.argv[1]
D = os.path.join(os.path.dirname(os.path.abspath(__file__)), tag)
t = json.load(open(os.path.join(D, "traffic.json")))
rows = []
for l in open(os.path.join(D, "frame_trace_1M_1024_cube.txt")).read().splitlines()[1:]:
    m = re.match(r'^(.*?)\s+([\d.]+)\s+([\d.]+)\s+(-?[\d.]+)\s+(\d+)\s*$', l)
    if not m:
        break
    name, us = m.group(1).strip(), float(m.group(3))
    kk = name if name in t["kernels"] else [k for k in t["kernels"] if k.startswith(name.split("<")[0])][0]
    rows.append((name, us, t["kernels"][kk]["hbm_bytes"], t["kernels"][kk]["valu_wave_insts"]))
out = ["One forward frame, 1 M Gaussians @ 1024x1024 (cube): every kernel against the two rooflines that can bind it.",
       f"duration: profiles/{tag}/frame_trace_1M_1024_cube.txt (one traced frame); HBM bytes and VALU wave-instructions per launch: traffic.json",
       "(PMC passes, hbm_bytes = 2 FETCH_SIZE + WRITE_SIZE); HBM peak 8 TB/s; VALU issue roofline = wave-instructions x 2.5 cycles /",
       "(1024 SIMDs x 2.4 GHz) -- the plain-VALU rate with >= 2 waves per SIMD (DESIGN.md section 4), a lower bound on the issue time.",
       "", f"{'kernel':44s} {'us':>7s} {'HBM MB':>8s} {'TB/s':>6s} {'of peak':>8s} {'VALU insts':>11s} {'issue us':>9s} {'of kernel':>9s}"]
tu = tb = 0.0
for name, us, b, vi in rows:
    issue = vi * 2.5 / (1024 * 2.4e9) * 1e6
    out.append(f"{name[:44]:44s} {us:7.2f} {b / 1e6:8.1f} {b / us / 1e6:6.2f} {b / us / 1e6 / 8:8.3f} {vi:11.3g} {issue:9.1f} {issue / us:9.2f}")
    tu += us; tb += b
out += [f"{'frame (%d launches, no gaps)' % len(rows):44s} {tu:7.2f} {tb / 1e6:8.1f} {tb / tu / 1e6:6.2f} {tb / tu / 1e6 / 8:8.3f}", "",
        "The preprocess kernel is the one near its bandwidth roofline (0.55 of the 8 TB/s peak, ~0.85 of what a mixed read / write stream",
        "reaches on this part).  The blend is bound by VALU issue (0.58 of the plain rate; its instructions cost 1.31 units on average,",
        "so ~0.75 of the SIMDs' real rate).  The six kernels between them are chains of dependent memory trips inside short workgroups",
        "on top of the floor of a dependent launch: 2.6 - 2.8 us for an empty kernel (profiles/r06/launch_floor_probe.txt; 8 launches =",
        "22 us of the frame), more behind a kernel that leaves dirty lines in the L2s."]
open(os.path.join(D, "frame_rooflines.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
