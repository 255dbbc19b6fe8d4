"""On the GPU box, after the rocprofv3 --pmc passes of tools/profile_run_r03.sh: per-kernel sums of every counter
of a pass -> <run>/pmc_<name>.json, then the pass's raw directory (a 19 MB database for the SQ passes) is removed, so
that what travels back stays under gpurun's 64 MiB.   python3 tools/pmc_extract.py gpurun_out/<run>"""
import glob, json, os, shutil, sqlite3, sys

run = sys.argv[1]


for d in sorted(glob.glob(os.path.join(run, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
        t = lambda p: [x for x in tabs if x.startswith(p)][0]
        q = f"""select p.name, s.kernel_name, d.id, sum(e.value) from {t('rocpd_pmc_event')} e
                join {t('rocpd_info_pmc')} p on e.pmc_id = p.id
                join {t('rocpd_kernel_dispatch')} d on e.event_id = d.event_id
                join {t('rocpd_info_kernel_symbol')} s on d.kernel_id = s.id group by 1, 2, 3"""
        for c, n, _, v in db.execute(q):
            out.setdefault(c, {}).setdefault(n, []).append(v)
        db.close()
    res = {c: {k: [sum(v), len(v)] for k, v in ks.items()} for c, ks in out.items()}  # per kernel name: [sum over dispatches, dispatches]
    json.dump(res, open(d + ".json", "w"))
    shutil.rmtree(d)
    print(os.path.basename(d), sorted(res))
