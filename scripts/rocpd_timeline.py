"""Timeline of the dispatches of a rocprofv3 rocpd database: start (µs from the first listed), duration, gap to the end of the previous dispatch, queue.
usage: python3 scripts/rocpd_timeline.py <results.db> [--between <marker substring>] [--skip N] [--count M]"""
import sqlite3
import sys


def tables(c, key):
    return [r[0] for r in c.execute("select name from sqlite_master where type='table'") if key in r[0]]


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def main():
    c = sqlite3.connect(sys.argv[1])
    kd, ks = tables(c, "rocpd_kernel_dispatch")[0], tables(c, "rocpd_info_kernel_symbol")[0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    qcol = "d.queue_id" if "queue_id" in cols else "0"
    where = ""
    key = arg("--between", None)
    if key:
        marks = [r[0] for r in c.execute(f"select d.start from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like ? order by d.start", ("%" + key + "%",))]
        where = " where d.start > %d and d.start < %d" % (marks[0], marks[1])
    rows = list(c.execute(f"select s.kernel_name, d.start, d.end, {qcol} from {kd} d join {ks} s on d.kernel_id=s.id{where} order by d.start"))
    skip, count = int(arg("--skip", "0")), int(arg("--count", "40"))
    rows = rows[skip:skip + count]
    t0, prev_end = rows[0][1], None
    for name, st, en, q in rows:
        gap = "" if prev_end is None else "%7.1f" % ((st - prev_end) / 1e3)
        print("%9.1f us  dur %8.1f  gap %7s  q %s  %s" % ((st - t0) / 1e3, (en - st) / 1e3, gap, q, name[:90]))
        prev_end = en if prev_end is None else max(prev_end, en)


if __name__ == "__main__":
    main()
