"""Summarise a rocprofv3 kernel trace CSV: busy time vs gaps for the last FRI commit in the trace."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last commit = from the last k_merkle_subtree<true> preceded by a gap > 100 us
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in rows]
starts = [i for i in range(1, len(ev)) if ev[i][0] - ev[i - 1][1] > 80_000]
i0 = starts[-1] if starts else 0
seg = ev[i0:]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0]
print(f"kernels {len(seg)}  span {span/1e3:.1f} us  busy {busy/1e3:.1f} us  gaps {(span-busy)/1e3:.1f} us")
for j, (s, e, name) in enumerate(seg):
    gap = s - seg[j - 1][1] if j else 0
    print(f"  +{gap/1e3:6.1f} us gap  {(e-s)/1e3:7.1f} us  {name}")
