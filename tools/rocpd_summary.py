#!/usr/bin/env python3
"""Dev tool: summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs as CSV for profiles/.

  rocpd_summary.py stats <run_results.db> <out.csv>   kernel name, calls, total/avg/min/max duration (ns), % (the --stats table)
  rocpd_summary.py pmc   <run_results.db> <out.csv>   kernel name, counter, dispatches, sum, mean per dispatch
"""
import csv, sqlite3, sys


def main():
    mode, path, out = sys.argv[1:4]
    db = sqlite3.connect(path)
    cur = db.cursor()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        if mode == "stats":
            rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc"))
            tot = sum(r[2] for r in rows) or 1
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
            for r in rows:
                w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), round(100.0 * r[2] / tot, 3)])
        else:
            rows = list(cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection group by kernel_name, counter_name order by sum(value) desc"))
            w.writerow(["Kernel", "Counter", "Dispatches", "Sum", "MeanPerDispatch"])
            for r in rows:
                w.writerow([r[0], r[1], r[2], r[3], r[4]])


if __name__ == "__main__":
    main()
