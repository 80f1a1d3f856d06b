"""Summarise a rocprofv3 rocpd SQLite database: per-kernel average duration and (if present) PMC counter sums per launch.
usage: python3 scripts/rocpd_summary.py <results.db> [--json out.json] [--between <substring of a marker kernel's name>]
--between: only the dispatches that START between the first and the second dispatch of a kernel whose name contains the substring (bench.py
--trace-markers brackets its timed region with two such launches: the averages are then those of the timed steps, not of the warm-up and the side measurements)."""
import json
import sqlite3
import sys


def tables(c, key):
    return [r[0] for r in c.execute("select name from sqlite_master where type='table'") if key in r[0]]


def main():
    c = sqlite3.connect(sys.argv[1])
    kd, ks = tables(c, "rocpd_kernel_dispatch")[0], tables(c, "rocpd_info_kernel_symbol")[0]
    out = {}
    where = ""
    if "--between" in sys.argv:
        key = sys.argv[sys.argv.index("--between") + 1]
        marks = [r[0] for r in c.execute(f"select d.start from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like ? order by d.start", ("%" + key + "%",))]
        if len(marks) < 2:
            print("no two marker dispatches matching %r in the trace" % key, file=sys.stderr)
            sys.exit(3)
        where = " where d.start > %d and d.start < %d and s.kernel_name not like '%%%s%%'" % (marks[0], marks[1], key.replace("'", ""))
    q = f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id{where} group by s.kernel_name order by 4 desc"
    for name, n, avg, tot in c.execute(q):
        out[name] = {"calls": n, "avg_us": avg / 1e3, "total_ms": tot / 1e6}
    pe, pi = tables(c, "rocpd_pmc_event"), tables(c, "rocpd_info_pmc")
    if pe and pi:
        q = (f"select s.kernel_name, p.name, sum(e.value), count(distinct d.id) from {pe[0]} e join {pi[0]} p on e.pmc_id=p.id "
             f"join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id{where} group by s.kernel_name, p.name")
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
