"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide
prescribes).  Usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps>
FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reads HALF of the bytes of wide coalesced reads, so the
corrected read figure is 2 x FETCH_SIZE (guide, HBM / rocprofv3 section)."""
import csv, sys, collections


def load(fp, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(fp) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"]
            acc[k][0] += 1
            acc[k][1] += float(row["Counter_Value"])
    return acc


def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 70 else name[:67] + "..."


fetch, write, steps = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
print("| kernel | launches | FETCH_SIZE (MB, raw) | fetch x2 (MB) | WRITE_SIZE (MB) | traffic = 2 x fetch + write (MB) |")
print("|---|---|---|---|---|---|")
tot_f = tot_w = 0.0
rows = []
for k, (n, v) in fetch.items():
    w = write.get(k, [n, 0.0])[1]
    tot_f += v; tot_w += w
    rows.append((v * 2 + w, k, n, v / n * 1024 / 1e6, w / max(write.get(k, [n])[0], 1) * 1024 / 1e6))
for _, k, n, f, w in sorted(rows, reverse=True)[:14]:
    print(f"| `{short(k)}` | {n} | {f:.1f} | {2 * f:.1f} | {w:.1f} | {2 * f + w:.1f} |")
print(f"\nAll kernels, {steps} steps: FETCH raw {tot_f * 1024 / 1e9:.2f} GB (x2 = {2 * tot_f * 1024 / 1e9:.2f} GB), WRITE {tot_w * 1024 / 1e9:.2f} GB "
      f"-> per step fetch x2 {2 * tot_f * 1024 / 1e9 / steps:.2f} GB, write {tot_w * 1024 / 1e9 / steps:.2f} GB.")
