"""Kernel statistics and a burst timeline out of a rocprofv3 result database (*_results.db, the rocpd SQLite schema of ROCm 7) -- for runs whose CSV conversion
did not finish.  usage: rocpd_stats.py results.db [--timeline N]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
names = {i: n for i, n in cur.execute(f"select id, kernel_name from {sym}")}
rows = list(cur.execute(f"select kernel_id, start, end, queue_id, grid_size_x, workgroup_size_x from {disp} order by start"))
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", ""))[:70]
agg = {}
for k, s, e, q, g, w in rows:
    a = agg.setdefault(names[k], [0, 0, 0]); a[0] += 1; a[1] += e - s; a[2] = max(a[2], e - s)
print("%-72s %6s %10s %10s %10s" % ("kernel", "calls", "total ms", "avg us", "max us"))
for n, (c, t, m) in sorted(agg.items(), key=lambda x: -x[1][1])[:28]:
    print("%-72s %6d %10.2f %10.1f %10.1f" % (short(n), c, t / 1e6, t / c / 1e3, m / 1e3))
if "--timeline" in sys.argv:
    nb = int(sys.argv[sys.argv.index("--timeline") + 1])
    # bursts: gaps of more than 3 ms between kernel activity
    bursts, curb, last_end = [], [], None
    for r in rows:
        if last_end is not None and r[1] - last_end > 3e6 and curb: bursts.append(curb); curb = []
        curb.append(r); last_end = max(last_end or 0, r[2])
    if curb: bursts.append(curb)
    for b in bursts[-nb:]:
        t0 = b[0][1]; span = max(r[2] for r in b) - t0
        print("burst: %d kernels, span %.1f ms" % (len(b), span / 1e6))
        for k, s, e, q, g, w in b:
            if e - s > 0.8e6: print("   + %6.1f  %6.1f ms  queue %-3d grid %8d wg %4d  %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, g, w, short(names[k])))
