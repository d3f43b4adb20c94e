"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel time stats and PMC counter sums.

    python tools/rocpd_summary.py stats  <results.db>  > profiles/<round>_kernel_stats.csv
    python tools/rocpd_summary.py pmc    <results.db>  > profiles/<round>_pmc_<counter>.csv
    python tools/rocpd_summary.py ranges <results.db>  > profiles/<round>_roctx_ranges.csv  (rocprofv3 --marker-trace with MEDPY_HIP_ROCTX=1: host time per named range)
    python tools/rocpd_summary.py timeline <results.db>  > profiles/<round>_timeline.csv    (every dispatch in order: start, gap to the one before, duration)
"""
import sqlite3
import sys


def main():
    mode, path = sys.argv[1], sys.argv[2]
    cur = sqlite3.connect(path).cursor()
    if mode == "stats":
        print("kernel,calls,total_us,avg_us,min_us,max_us,percent")
        rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                           "group by name order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        for n, c, s, a, mn, mx in rows:
            print('"%s",%d,%.3f,%.3f,%.3f,%.3f,%.2f' % (n, c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    elif mode == "ranges":
        import collections
        import json
        acc = collections.OrderedDict()
        for ext, dur in cur.execute("select extdata, duration from regions where category like 'MARKER%' order by start"):
            try:
                name = json.loads(ext).get("message", "?")
            except ValueError:
                name = "?"
            a = acc.setdefault(name, [0, 0])
            a[0] += 1
            a[1] += dur
        print("range,calls,total_us,avg_us")
        for name, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            print('"%s",%d,%.1f,%.1f' % (name, c, d / 1e3, d / 1e3 / c))
    elif mode == "timeline":
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
        st, en = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
        rows = cur.execute("select name, %s, %s from kernels order by %s" % (st, en, st)).fetchall()
        print("start_us,gap_us,duration_us,kernel")
        t0, last_end = (rows[0][1] if rows else 0), None
        for n, a, b in rows:
            print("%.2f,%.2f,%.2f,%s" % ((a - t0) / 1e3, 0.0 if last_end is None else (a - last_end) / 1e3, (b - a) / 1e3, n.split("(")[0].replace("void ", "")))
            last_end = b
    elif mode == "json":
        # python tools/rocpd_summary.py json <fetch.db> <write.db> > profiles/pmc_discharge.json
        import json
        kern = sys.argv[4] if len(sys.argv) > 4 else "k_discharge_w"  # (json <fetch.db> <write.db> k26_discharge: the 26-neighbourhood kernel of bench.py --config 3)
        out = {"kernel": kern}
        for key, db in (("fetch_kib_per_launch", sys.argv[2]), ("write_kib_per_launch", sys.argv[3])):
            c = sqlite3.connect(db).cursor()
            # (the discharge kernels are templates: "void k_discharge_w<1>(MgcLattice, ...)"; every instance counts)
            n, v = c.execute("select count(*), avg(value) from counters_collection where kernel_name like ? or kernel_name like ? or kernel_name like ?",
                             (kern + "(%", "void " + kern + "<%", "void " + kern + "(%")).fetchone()
            out[key] = v
            out[key.replace("kib_per_launch", "launches")] = n
        import hashlib, os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        h = hashlib.sha256()
        d = os.path.join(root, "medpy_amd", "csrc")
        for f in sorted(os.listdir(d)):
            h.update(open(os.path.join(d, f), "rb").read())
        out["kernel_sources"] = h.hexdigest()[:16]  # bench.py only quotes these counters for the kernels they were taken with
        print(json.dumps(out))
    else:
        print("kernel,counter,dispatches,sum_value,avg_value_per_dispatch,avg_duration_us")
        rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
                           "group by kernel_name, counter_name order by sum(value) desc").fetchall()
        for n, c, k, s, a, d in rows:
            print('"%s",%s,%d,%.3f,%.3f,%.3f' % (n, c, k, s, a, d / 1e3))


if __name__ == "__main__":
    main()
