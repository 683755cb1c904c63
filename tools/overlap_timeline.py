"""Rank 0's GPU timeline around the sums of one step, from tools/overlap_trace.sh's rocprofv3 kernel trace:
   python3 tools/overlap_timeline.py gpurun_out/ovt_<tag> profiles/<tag>_overlap_timeline.json [anchor-kernel-prefix]
Takes the LAST step of the run: from the launch that follows the search (blob_dedup_kernel, which also sorts the
workgroups into interior / boundary) to the last output kernel after the last blob pass.  Host<->device copies of the halo appear as
__amd_rocclr_copyBuffer launches (the gloo rehearsal moves the halo through host memory)."""
import csv
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(os.path.join(src, "prof", "r0_kernel_trace.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda nm: nm.split("(")[0].replace("void ", "")[:48]
names = [short(r["Kernel_Name"]) for r in rows]
anchor = sys.argv[3] if len(sys.argv) > 3 else "blob_dedup_kernel"
starts = [i for i, n in enumerate(names) if n.startswith(anchor)]
i0 = starts[-1]
blob = [i for i, n in enumerate(names) if n.startswith("blob_") and i >= i0 and not n.startswith(("blob_dedup", "blob_split", "blob_count"))]
i1 = blob[-1]
while i1 + 1 < len(names) and names[i1 + 1].startswith(("dev_to_caller", "scatter_rows")):
    i1 += 1
t0 = int(rows[i0]["Start_Timestamp"])
tl = []
for r, n in zip(rows[i0:i1 + 1], names[i0:i1 + 1]):
    tl.append({"kernel": n, "start_us": round((int(r["Start_Timestamp"]) - t0) / 1e3, 1),
               "dur_us": round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1),
               "workgroups": int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)})
out = {"source": "rocprofv3 --kernel-trace of rank 0, two ranks sharing one MI355X over gloo (tools/overlap_trace.sh)",
       "note": "blob_* pass kernels appear twice per pass when the split is on: interior workgroups first (launched while the halo "
               "phase is in flight: between the copyBuffer that takes the sent rows to the host and the one that brings the "
               "received rows back), boundary workgroups after it",
       "timeline": tl}
json.dump(out, open(dst, "w"), indent=1)
for e in tl:
    print("%9.1f %8.1f  %s" % (e["start_us"], e["dur_us"], e["kernel"]))
