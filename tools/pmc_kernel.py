"""Mean per-dispatch value of every counter for kernels whose name contains a pattern (rocprofv3 --pmc CSV)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for fp in sys.argv[2:]:
    with open(fp) as f:
        for row in csv.DictReader(f):
            if sys.argv[1] in row["Kernel_Name"]:
                a = acc[row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"{k:36s} {v / n:16.0f}   (n={n})")
