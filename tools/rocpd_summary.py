"""Summarise a rocprofv3 rocpd sqlite database: one row per (kernel, grid size) -- launches of different sizes are different
workloads and must not share an average -- with count / avg / min / max duration, and the PMC sums per call when counters were
collected.  `--by-kernel` merges the grids of a kernel again (the old behaviour)."""
import collections
import sqlite3
import sys


def table(cur, prefix):
    r = cur.execute("select name from sqlite_master where type='table' and name like ?", (prefix + '%',)).fetchall()
    return r[0][0] if r else None


def main(path, by_kernel=False):
    con = sqlite3.connect(path)
    cur = con.cursor()
    kd, ks = table(cur, 'rocpd_kernel_dispatch'), table(cur, 'rocpd_info_kernel_symbol')
    cols = [r[1] for r in cur.execute(f"pragma table_info('{ks}')")]
    namecol = 'kernel_name' if 'kernel_name' in cols else ('display_name' if 'display_name' in cols else cols[-2])
    names = {r[0]: r[1] for r in cur.execute(f"select id, {namecol} from '{ks}'")}
    stats = collections.OrderedDict()
    ev2k = {}
    for kid, s, e, evid, gx, wx, lds in cur.execute(f"select kernel_id, start, end, event_id, grid_size_x, workgroup_size_x, group_segment_size from '{kd}'"):
        n = str(names.get(kid, kid))
        short = n.split('(')[0][-60:]
        key = short if by_kernel else (short, gx)
        d = stats.setdefault(key, {'name': short, 'calls': 0, 'total_ns': 0, 'min': 1 << 62, 'max': 0, 'grid': gx, 'wg': wx, 'lds': lds,
                                   'pmc': collections.defaultdict(float)})
        d['calls'] += 1
        d['total_ns'] += e - s
        d['min'] = min(d['min'], e - s)
        d['max'] = max(d['max'], e - s)
        ev2k[evid] = key
    pe, pi = table(cur, 'rocpd_pmc_event'), table(cur, 'rocpd_info_pmc')
    if pe and pi:
        pname = {r[0]: r[1] for r in cur.execute(f"select id, name from '{pi}'")}
        for evid, pid, val in cur.execute(f"select event_id, pmc_id, value from '{pe}'"):
            if evid in ev2k:
                stats[ev2k[evid]]['pmc'][pname.get(pid, str(pid))] += val
    print(f"{'kernel':62s} {'grid(thr)':>10s} {'wg':>5s} {'calls':>6s} {'avg_us':>12s} {'min_us':>11s} {'max_us':>11s} {'total_ms':>10s} {'lds':>7s}")
    for _, d in sorted(stats.items(), key=lambda kv: -kv[1]['total_ns']):
        print(f"{d['name']:62s} {d['grid']:10d} {d['wg']:5d} {d['calls']:6d} {d['total_ns'] / d['calls'] / 1e3:12.2f} {d['min'] / 1e3:11.2f} "
              f"{d['max'] / 1e3:11.2f} {d['total_ns'] / 1e6:10.3f} {d['lds']:7d}")
        for pn, v in sorted(d['pmc'].items()):
            print(f"      {pn:34s} {v / d['calls']:18.1f}  (per call)")


def timeline(path, substrings, last=400):
    """dispatches whose kernel name contains one of `substrings`, in start order: start (ms from the first), duration, gap to the previous end"""
    con = sqlite3.connect(path)
    cur = con.cursor()
    kd, ks = table(cur, 'rocpd_kernel_dispatch'), table(cur, 'rocpd_info_kernel_symbol')
    cols = [r[1] for r in cur.execute(f"pragma table_info('{ks}')")]
    namecol = 'kernel_name' if 'kernel_name' in cols else ('display_name' if 'display_name' in cols else cols[-2])
    names = {r[0]: r[1] for r in cur.execute(f"select id, {namecol} from '{ks}'")}
    dcols = [r[1] for r in cur.execute(f"pragma table_info('{kd}')")]
    qcol = 'queue_id' if 'queue_id' in dcols else ('stream_id' if 'stream_id' in dcols else '0')  # which HIP stream's queue: overlap between streams
    rows = [(s, e, str(names.get(k, k)).split('(')[0][-48:], gx, q) for k, s, e, gx, q in cur.execute(f"select kernel_id, start, end, grid_size_x, {qcol} from '{kd}'")]
    rows = sorted(r for r in rows if any(x in r[2] for x in substrings))[-last:]
    t0, prev = rows[0][0], rows[0][0]
    for s, e, n, gx, q in rows:
        print(f"{(s - t0) / 1e6:10.3f} ms  dur {(e - s) / 1e3:9.1f} us  gap {(s - prev) / 1e3:8.1f} us  q{q}  {n:48s} grid {gx}")
        prev = max(prev, e)


if __name__ == '__main__':
    if '--timeline' in sys.argv[2:]:
        i = sys.argv.index('--timeline')
        timeline(sys.argv[1], sys.argv[i + 1].split(','), int(sys.argv[i + 2]) if len(sys.argv) > i + 2 else 400)
    else:
        main(sys.argv[1], by_kernel='--by-kernel' in sys.argv[2:])
