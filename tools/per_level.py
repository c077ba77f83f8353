"""summarise gpurun_out/ngp_prof/ngp_kernel_trace.csv for the per-level launches of the encode backward"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "encode_bwd_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for k, r in enumerate(rows):
    per[k % 16].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for l in range(16):
    v = per[l][len(per[l]) // 2:]
    print(f"level {l:2d}: mean {sum(v)/len(v):8.1f} us  max {max(v):8.1f}")
