"""rocprofv3 --pmc databases of `bench.py` (one pass with FETCH_SIZE, one with WRITE_SIZE) -> profiles/r02_match_pmc.json: HBM bytes
per launch of match_fused_kernel.  FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: on gfx950 it reports half of the bytes of
wide coalesced reads); FETCH_SIZE / WRITE_SIZE are in KiB... the unit is taken from the counter description: kilobytes.
usage: pmc_to_json.py <fetch.db> <write.db> <pairs_per_launch> <out.json>"""
import collections, json, sqlite3, sys


def per_call(path, kernel_substr, counter):
    con = sqlite3.connect(path)
    cur = con.cursor()

    def table(prefix):
        r = cur.execute("select name from sqlite_master where type='table' and name like ?", (prefix + '%',)).fetchall()
        return r[0][0] if r else None

    kd, ks, pe, pi = table('rocpd_kernel_dispatch'), table('rocpd_info_kernel_symbol'), table('rocpd_pmc_event'), table('rocpd_info_pmc')
    cols = [r[1] for r in cur.execute(f"pragma table_info('{ks}')")]
    namecol = 'kernel_name' if 'kernel_name' in cols else 'display_name'
    names = {r[0]: r[1] for r in cur.execute(f"select id, {namecol} from '{ks}'")}
    ev = {evid: names.get(kid, '') for kid, evid in cur.execute(f"select kernel_id, event_id from '{kd}'")}
    pname = {r[0]: r[1] for r in cur.execute(f"select id, name from '{pi}'")}
    tot, calls = 0.0, set()
    for evid, pid, val in cur.execute(f"select event_id, pmc_id, value from '{pe}'"):
        if kernel_substr in str(ev.get(evid, '')) and pname.get(pid) == counter:
            tot += val
            calls.add(evid)
    return tot / max(1, len(calls)), len(calls)


def main_ba(generic=False):
    """usage: pmc_to_json.py --ba <fetch.db> <write.db> <observations> <out.json>: one Schur mat-vec = schur_point_coop_kernel<0> +
    schur_shot_kernel (per-call averages added); --ba-generic: the generic rows' pair, gen_schur_point_kernel<2, 0> + gen_schur_shot_kernel"""
    fetch_db, write_db, nobs, out = sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
    f = w = 0.0
    calls = []
    for k in (("gen_schur_point_kernelILi2ELi0E", "gen_schur_shot_kernel") if generic else ("23schur_point_coop_kernelILi0E", "17schur_shot_kernel")):
        fk, nf = per_call(fetch_db, k, 'FETCH_SIZE')
        wk, nw = per_call(write_db, k, 'WRITE_SIZE')
        f += fk
        w += wk
        calls.append([k, nf, nw, fk, wk])
    json.dump({"kernel": "schur mat-vec, generic rows (gen_schur_point_kernel<2, 0> + gen_schur_shot_kernel)" if generic else
               "schur mat-vec (schur_point_coop_kernel<0> + schur_shot_kernel)", "units_per_launch": nobs, "per_kernel": calls,
               "fetch_size_kb_raw": f, "write_size_kb_raw": w, "fetch_bytes_corrected": 2.0 * f * 1024.0, "write_bytes": w * 1024.0,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_passes.sh) of `python tools/prof_ba.py 5000 500000 10 5" +
                         (" general" if generic else "") + "`; FETCH_SIZE x 2 per "
                         "MI355X_MICROARCH.md (gfx950 reports 64 B per 128 B request); counter unit KiB"}, open(out, 'w'), indent=1)
    print(open(out).read())


def main():
    if sys.argv[1] in ("--ba", "--ba-generic"):
        return main_ba(sys.argv[1] == "--ba-generic")
    fetch_db, write_db, ppl, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
    f, nf = per_call(fetch_db, 'match_fused_kernel', 'FETCH_SIZE')
    w, nw = per_call(write_db, 'match_fused_kernel', 'WRITE_SIZE')
    json.dump({"kernel": "match_fused_kernel", "pairs_per_launch": ppl, "launches_sampled": [nf, nw],
               "fetch_size_kb_raw": f, "write_size_kb_raw": w,
               "fetch_bytes_corrected": 2.0 * f * 1024.0, "write_bytes": w * 1024.0,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_passes.sh) of `python bench.py --headline-only --no-cpu-baseline --steps 1 "
                         "--warmup 0`; FETCH_SIZE x 2 per MI355X_MICROARCH.md (gfx950 reports 64 B per 128 B request); "
                         "counter unit KiB"}, open(out, 'w'), indent=1)
    print(open(out).read())


if __name__ == '__main__':
    main()
