"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count / avg duration, and PMC sums."""
import sqlite3, sys, collections, json

def table(cur, prefix):
    r = cur.execute("select name from sqlite_master where type='table' and name like ?", (prefix + '%',)).fetchall()
    return r[0][0] if r else None

def main(path):
    con = sqlite3.connect(path); cur = con.cursor()
    kd, ks, st = table(cur, 'rocpd_kernel_dispatch'), table(cur, 'rocpd_info_kernel_symbol'), table(cur, 'rocpd_string')
    cols = [r[1] for r in cur.execute(f"pragma table_info('{ks}')")]
    namecol = 'kernel_name' if 'kernel_name' in cols else ('display_name' if 'display_name' in cols else cols[-2])
    names = {r[0]: r[1] for r in cur.execute(f"select id, {namecol} from '{ks}'")}
    stats = collections.OrderedDict()
    ev2k = {}
    for kid, s, e, evid, gx, wx, lds in cur.execute(f"select kernel_id, start, end, event_id, grid_size_x, workgroup_size_x, group_segment_size from '{kd}'"):
        n = str(names.get(kid, kid))
        short = n.split('(')[0][-60:]
        d = stats.setdefault(short, {'calls': 0, 'total_ns': 0, 'grid': gx, 'wg': wx, 'lds': lds, 'pmc': collections.defaultdict(float)})
        d['calls'] += 1; d['total_ns'] += e - s
        ev2k[evid] = short
    pe, pi = table(cur, 'rocpd_pmc_event'), table(cur, 'rocpd_info_pmc')
    if pe and pi:
        pname = {r[0]: r[1] for r in cur.execute(f"select id, name from '{pi}'")}
        for evid, pid, val in cur.execute(f"select event_id, pmc_id, value from '{pe}'"):
            if evid in ev2k:
                stats[ev2k[evid]]['pmc'][pname.get(pid, str(pid))] += val
    print(f"{'kernel':62s} {'calls':>6s} {'avg_us':>12s} {'total_ms':>10s} {'grid':>9s} {'lds':>7s}")
    for k, d in sorted(stats.items(), key=lambda kv: -kv[1]['total_ns']):
        print(f"{k:62s} {d['calls']:6d} {d['total_ns']/d['calls']/1e3:12.2f} {d['total_ns']/1e6:10.3f} {d['grid']:9d} {d['lds']:7d}")
        for pn, v in sorted(d['pmc'].items()):
            print(f"      {pn:34s} {v/d['calls']:18.1f}  (per call)")

if __name__ == '__main__':
    main(sys.argv[1])
