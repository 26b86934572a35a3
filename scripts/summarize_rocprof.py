"""Condense a rocprofv3 --stats kernel_stats.csv into a short markdown table (kernel names
shortened) for profiles/."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([\w:]+(?:<[^()]*?>)?)\(", n)
    n = m.group(1) if m else n
    return n[:90]
print("| kernel | calls | avg us | min us | max us | total ms | % |")
print("|---|---:|---:|---:|---:|---:|---:|")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("| %s | %s | %.1f | %.1f | %.1f | %.2f | %.2f |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
          float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
