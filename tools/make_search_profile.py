#!/usr/bin/env python3
"""profiles/latest_<tag>_profile.json from a tools/pmc_profile.sh summary (per-launch counter means) and the kernel
stats CSV of the same build:  tools/make_search_profile.py <prefix> <n> <k> [tag]   (reads gpurun_out/<prefix>_*;
tag = bench.py's workload_tag, e.g. polytrope, sedov, uniform_cube_loop, two_phase_loop_species_drag; default polytrope).
Also the whole step's counter traffic (sum over every kernel of per-launch bytes x launches per step).
HBM-side bytes as MI355X_MICROARCH.md prescribes for gfx950: 2 x FETCH_SIZE (KB; the counter reports half of wide
streaming reads) + WRITE_SIZE (KB); 4-byte-per-lane stores are uncalibrated there, so the figure is an upper estimate."""
import csv, json, sys
prefix, n, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tag = sys.argv[4] if len(sys.argv) > 4 else "polytrope"
pmc = json.load(open("gpurun_out/%s_pmc_per_launch.json" % prefix))
ks = {r["Name"]: r for r in csv.DictReader(open("gpurun_out/%s_kernel_stats.csv" % prefix))}
def avg_us(sub):
    return [float(r["AverageNs"]) / 1e3 for nme, r in ks.items() if sub in nme][0]
grp = pmc["knn_group_kernel"]
lst = [v for nme, v in pmc.items() if nme.startswith("knn_kernel<0, 1, 1")][0]      # list mode (LIST = 1)
# whole step: every kernel's HBM-side bytes per launch x its launches per step (launch counts from the kernel-stats CSV
# of a run of `steps_in_csv` steps: the grouped kernel runs once per hinted step)
steps_in_csv = max(1.0, float(ks[[nme for nme in ks if "knn_group_kernel" in nme][0]]["Calls"]))
step_traffic = 0.0
for nme, c in pmc.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    calls = [float(r["Calls"]) for kn, r in ks.items() if kn.replace("void ", "").split("(")[0].strip() == nme.split("(")[0].strip()]
    if calls:
        step_traffic += (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 * (calls[0] / steps_in_csv)
traffic = sum((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 for c in (grp, lst))
cycles = grp["GRBM_GUI_ACTIVE"] / 8.0                       # per XCD
out = {
    "n": n, "k": k, "source": "profiles/%s_pmc_per_launch.json + profiles/%s_kernel_stats.csv" % (prefix, prefix),
    "traffic_bytes_per_launch": traffic,
    "traffic_note": "2 x FETCH_SIZE + WRITE_SIZE of knn_group_kernel and of the list-mode knn_kernel<0,1,1>, per step",
    "valu_issue_frac": grp["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * cycles),
    "valu_wave_instr_per_query": (grp["SQ_INSTS_VALU"] + lst["SQ_INSTS_VALU"]) / n,
    "l2_hit_rate": grp["TCC_HIT_sum"] / (grp["TCC_HIT_sum"] + grp["TCC_MISS_sum"]),
    "wave_wait_frac": grp["SQ_WAIT_ANY"] / grp["SQ_WAVE_CYCLES"],
    "kernel_us": {"knn_group_kernel": avg_us("knn_group_kernel"), "knn_kernel<0,1,1>": avg_us("knn_kernel<0, 1, 1")},
    "tag": tag,
    "step_traffic_bytes": step_traffic,
    "step_traffic_note": "sum over all kernels of (2 x FETCH_SIZE + WRITE_SIZE) per launch x launches per step",
}
json.dump(out, open("profiles/latest_%s_profile.json" % tag, "w"), indent=1)
if tag == "polytrope":
    json.dump(out, open("profiles/latest_search_profile.json", "w"), indent=1)
print(json.dumps(out, indent=1))
