import sqlite3, sys, glob, collections
for f in glob.glob(f"gpurun_out/{sys.argv[1]}/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    def t(prefix):
        return [x for x in tabs if x.startswith(prefix)][0]
    ev, pmc, disp, sym = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    q = f"""select s.kernel_name, p.name, avg(e.value), count(*) from {ev} e
            join {pmc} p on e.pmc_id = p.id join {disp} d on e.event_id = d.event_id
            join {sym} s on d.kernel_id = s.id group by 1,2"""
    for k, c, v, n in db.execute(q):
        if len(sys.argv) > 2 and sys.argv[2] not in k: continue
        print(f"{k.split('(')[0][:40]:40s} {c:28s} {v:16.1f} (n={n})")
