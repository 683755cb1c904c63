#!/usr/bin/env python3
"""profiles/latest_search_profile.json from a tools/pmc_profile.sh summary (per-launch counter means) and the kernel
stats CSV of the same build:  tools/make_search_profile.py <prefix> <n> <k>   (reads gpurun_out/<prefix>_*).
HBM-side bytes as MI355X_MICROARCH.md prescribes for gfx950: 2 x FETCH_SIZE (KB; the counter reports half of wide
streaming reads) + WRITE_SIZE (KB); 4-byte-per-lane stores are uncalibrated there, so the figure is an upper estimate."""
import csv, json, sys
prefix, n, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
pmc = json.load(open("gpurun_out/%s_pmc_per_launch.json" % prefix))
ks = {r["Name"]: r for r in csv.DictReader(open("gpurun_out/%s_kernel_stats.csv" % prefix))}
def avg_us(sub):
    return [float(r["AverageNs"]) / 1e3 for nme, r in ks.items() if sub in nme][0]
grp = pmc["knn_group_kernel"]
lst = [v for nme, v in pmc.items() if nme.startswith("knn_kernel<0, 1, 1")][0]      # list mode (LIST = 1)
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
}
json.dump(out, open("profiles/latest_search_profile.json", "w"), indent=1)
print(json.dumps(out, indent=1))
