"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into
HBM-side bytes per input byte for every pipeline stage, and merge them into
profiles/traffic.json (read by bench.py for roofline.traffic).

  python tools/pmc_traffic.py <fetch_run_dir> <write_run_dir> <kind/params> <input_bytes_per_launch>

Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on
gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled;
WRITE_SIZE is taken as reported.  Infinity-cache hits are included in both.
"""
import glob
import json
import os
import sqlite3
import sys

STAGES = {"k1_chain_build": ["k1_chain_build", "k1x_chain_build", "k1p_partition", "k1c_consume"],
          "k3_parse": ["k3a_spec", "k3a_fast", "k3b_stitch", "k3c_validate"],
          "k4_compact": ["k4_scan_counts", "k4_compact"],
          # (the histogram-only instantiation: a run with the gpu_encode leg also holds k5_histograms<true>)
          "k5_histograms": ["k5_histogramsILb0", "k5_histograms<false>", "k5b_long_literals"]}


def per_dispatch(run_dir, counter):
    j = run_dir.rstrip("/") + ".json"  # made on the GPU box by tools/pmc_extract.py: {counter: {kernel: [sum, dispatches]}}
    if os.path.exists(j):
        return {k: v[0] / v[1] for k, v in json.load(open(j)).get(counter, {}).items()}
    out = {}
    for f in glob.glob(os.path.join(run_dir, "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
        t = lambda p: [x for x in tabs if x.startswith(p)][0]
        q = f"""select s.kernel_name, d.id, sum(e.value) from {t('rocpd_pmc_event')} e
                join {t('rocpd_info_pmc')} p on e.pmc_id = p.id
                join {t('rocpd_kernel_dispatch')} d on e.event_id = d.event_id
                join {t('rocpd_info_kernel_symbol')} s on d.kernel_id = s.id
                where p.name = ? group by 1, 2"""
        for name, _, v in db.execute(q, (counter,)):
            out.setdefault(name, []).append(v)
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    fetch_dir, write_dir, key, nbytes = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    fetch = per_dispatch(fetch_dir, "FETCH_SIZE")
    write = per_dispatch(write_dir, "WRITE_SIZE")
    entry, detail = {}, {}
    for stage, kernels in STAGES.items():
        fb = sum(v for k, v in fetch.items() if any(x in k for x in kernels)) * 1024 * 2
        wb = sum(v for k, v in write.items() if any(x in k for x in kernels)) * 1024
        entry[stage] = round((fb + wb) / nbytes, 4)
        detail[stage] = {"fetch_bytes_per_input_byte": round(fb / nbytes, 4),
                         "write_bytes_per_input_byte": round(wb / nbytes, 4)}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = entry
    data[key + "/detail"] = detail
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    print(key, json.dumps(detail))


if __name__ == "__main__":
    main()
