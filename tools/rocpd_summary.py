"""Summaries of a rocprofv3 rocpd database (the only output format of this image's rocprofv3):
   python tools/rocpd_summary.py stats <db> <out.csv>      per-kernel calls / total / average / min / max ns
   python tools/rocpd_summary.py pmc <db> <counter>        per-dispatch sums of one counter, per kernel"""
import collections
import csv
import sqlite3
import sys


def table(c, prefix):
    return [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like ?", (prefix + "%",))][0]


def main():
    mode, path = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(path)
    kd, ks = table(c, "rocpd_kernel_dispatch"), table(c, "rocpd_info_kernel_symbol")
    if mode == "stats":
        # one row per (kernel, grid): the bench's timed launches (all node alignments in one grid) are told
        # apart from the smaller per-level launches of the untimed tree walk
        rows = collections.defaultdict(list)
        for name, gx, wx, start, end in c.execute(
                f"select k.kernel_name, d.grid_size_x, d.workgroup_size_x, d.start, d.end from {kd} d join {ks} k on d.kernel_id = k.id"):
            rows["%s [grid %d x wg %d]" % (name, gx // max(wx, 1), wx)].append(end - start)
        total = sum(sum(v) for v in rows.values())
        with open(sys.argv[3], "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([name, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / total, min(v), max(v)])
    else:
        pe, ip = table(c, "rocpd_pmc_event"), table(c, "rocpd_info_pmc")
        per = collections.defaultdict(float)
        q = (f"select k.kernel_name, d.grid_size_x, d.workgroup_size_x, d.id, e.value from {pe} e join {ip} p on e.pmc_id = p.id "
             f"join {kd} d on e.event_id = d.event_id join {ks} k on d.kernel_id = k.id where p.name = ?")
        for name, gx, wx, did, value in c.execute(q, (sys.argv[3],)):
            per[("%s [grid %d x wg %d]" % (name, gx // max(wx, 1), wx), did)] += value
        by = collections.defaultdict(list)
        for (name, did), v in sorted(per.items(), key=lambda kv: kv[0][1]):
            by[name].append(v)
        for name, v in by.items():
            print(name, [round(x, 3) for x in v])


if __name__ == "__main__":
    main()
