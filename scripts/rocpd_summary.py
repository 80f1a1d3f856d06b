"""Summarise a rocprofv3 rocpd SQLite database: per-kernel average duration and (if present) PMC counter sums per launch.
usage: python3 scripts/rocpd_summary.py <results.db> [--json out.json]"""
import json
import sqlite3
import sys


def tables(c, key):
    return [r[0] for r in c.execute("select name from sqlite_master where type='table'") if key in r[0]]


def main():
    c = sqlite3.connect(sys.argv[1])
    kd, ks = tables(c, "rocpd_kernel_dispatch")[0], tables(c, "rocpd_info_kernel_symbol")[0]
    out = {}
    q = f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc"
    for name, n, avg, tot in c.execute(q):
        out[name] = {"calls": n, "avg_us": avg / 1e3, "total_ms": tot / 1e6}
    pe, pi = tables(c, "rocpd_pmc_event"), tables(c, "rocpd_info_pmc")
    if pe and pi:
        q = (f"select s.kernel_name, p.name, sum(e.value), count(distinct d.id) from {pe[0]} e join {pi[0]} p on e.pmc_id=p.id "
             f"join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by s.kernel_name, p.name")
        try:
            for name, ctr, val, n in c.execute(q):
                out[name].setdefault("pmc_per_launch", {})[ctr] = val / max(n, 1)
        except sqlite3.Error as e:  # schema differs between ROCm releases
            print("pmc query failed:", e, file=sys.stderr)
    for name, v in out.items():
        print("%-110s calls %4d avg %10.1f us" % (name[:110], v["calls"], v["avg_us"]))
        for k, x in sorted(v.get("pmc_per_launch", {}).items()):
            print("      %-28s %.4g" % (k, x))
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
