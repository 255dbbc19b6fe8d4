"""profiles/r03_* from one measurement run directory made by tools/profile_run_r03.sh:
   python tools/make_profile_summary_r03.py gpurun_out/r03a r03"""
import csv, glob, json, os, re, shutil, sqlite3, subprocess, sys

run, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(root, "profiles", *a)
N1K = 1024 * 1048576  # input bytes of the 1024-chunk launches the PMC passes and trace1k ran


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def short(n):
    m = re.search(r"_ZN3m4g(\d+)([A-Za-z0-9_]+)", n)
    if m:
        return m.group(2)[:int(m.group(1))]
    return n.split("(")[0].replace("void ", "").replace("m4g::", "").split("<")[0]


def per_kernel(sub, counter):
    j = os.path.join(run, sub + ".json")  # made on the GPU box by tools/pmc_extract.py (same query as below)
    if os.path.exists(j):
        acc = {}
        for n, (tot, cnt) in json.load(open(j)).get(counter, {}).items():
            a = acc.setdefault(short(n), [0.0, 0])
            a[0] += tot
            a[1] += cnt
        return {k: v[0] / v[1] for k, v in acc.items()}
    out = {}
    for f in glob.glob(os.path.join(run, sub, "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
        t = lambda p: [x for x in tabs if x.startswith(p)][0]
        q = f"""select s.kernel_name, d.id, sum(e.value) from {t('rocpd_pmc_event')} e
                join {t('rocpd_info_pmc')} p on e.pmc_id = p.id
                join {t('rocpd_kernel_dispatch')} d on e.event_id = d.event_id
                join {t('rocpd_info_kernel_symbol')} s on d.kernel_id = s.id where p.name = ? group by 1, 2"""
        for n, _, v in db.execute(q, (counter,)):
            out.setdefault(short(n), []).append(v)
    return {k: sum(v) / len(v) for k, v in out.items()}


def stats_of(sub):
    f = glob.glob(os.path.join(run, sub, "**", "*kernel_stats.csv"), recursive=True)[0]
    return f, {short(r["Name"]): float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(f))}


stats_file, ms4k = stats_of("trace")
_, ms1k = stats_of("trace1k")
shutil.copy(stats_file, P(f"{tag}_kernel_stats.csv"))
try:
    shutil.copy(stats_of("trace_legs")[0], P(f"{tag}_kernel_stats_legs.csv"))
except Exception:
    pass
try:  # the default (overlapped) form: half-size launches next to each other, durations not comparable with stage_ms
    shutil.copy(stats_of("trace_ov")[0], P(f"{tag}_kernel_stats_overlapped.csv"))
except Exception:
    pass
b = last_json(os.path.join(run, "bench.json"))
bl, bd = last_json(os.path.join(run, "bench_log.json")), last_json(os.path.join(run, "bench_D.json"))
for name, d in ((f"{tag}_bench.json", b), (f"{tag}_bench_log.json", bl), (f"{tag}_bench_D.json", bd)):
    json.dump(d, open(P(name), "w"))
for f in ("kind_probe.log", "stream_probe.log", "hostbatch_probe.log", "writer_probe.log", "compress_probe.log",
          "bargain_probe.log", "bargain_stats.log", "v2_probe.log"):
    if os.path.exists(os.path.join(run, f)):
        shutil.copy(os.path.join(run, f), P(f"{tag}_{f}"))
for key, fd, wd in (("text/G", "pmc_fetch", "pmc_write"), ("text/D", "pmc_fetch_D", "pmc_write_D"),
                    ("log/G", "pmc_fetch_log", "pmc_write_log")):
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "pmc_traffic.py"), os.path.join(run, fd),
                           os.path.join(run, wd), key, str(N1K)])

# ---- per-kernel counter table (north_star: achieved HBM GB/s, LDS bank conflicts, VALU busy) ----
F, W = per_kernel("pmc_fetch", "FETCH_SIZE"), per_kernel("pmc_write", "WRITE_SIZE")
C = {}
for sub, names in (("pmc_sq1", ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_LDS",
                                "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY"]),
                   ("pmc_sq2", ["SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                                "SQ_INSTS_SALU", "SQ_WAVES", "SQ_ACTIVE_INST_SCA"])):
    for n in names:
        for k, v in per_kernel(sub, n).items():
            C.setdefault(k, {})[n] = v
kernels = ["k1p_partition", "k1c_consume", "k3a_fast", "k3a_spec", "k3b_stitch", "k3c_validate", "k4_compact", "k5_histograms", "k6_emit_bodies"]
T = [f"# Round {tag[1:]}: per-kernel counters (1 x MI355X, 1024 x 1 MiB text chunks, params G, per launch)\n",
     "Separate `rocprofv3 --pmc` passes (`tools/profile_run_r03.sh`: FETCH_SIZE | WRITE_SIZE | 8 SQ counters | 8 SQ counters)",
     "over `python3 bench.py --chunks 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --verify-chunks 0`;",
     "durations from a `--kernel-trace --stats` run of the same command.  FETCH_SIZE is doubled (gfx950 tallies 128-byte",
     "requests at 64 bytes, MI355X_MICROARCH.md); both sizes are KiB.  The SQ_* cycle counters are sums over all waves, so",
     "the table gives ratios: the share of a wave's lifetime (SQ_WAVE_CYCLES) in which it issues VALU / LDS / any",
     "instruction (SQ_ACTIVE_INST_*) or waits for one (SQ_WAIT_INST_ANY), and LDS bank-conflict cycles per LDS busy cycle",
     "(SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).  With 4 to 8 waves per SIMD taking turns, a VALU share of 1 / waves means",
     "the SIMD's VALU is busy all the time.\n",
     "| kernel | ms | HBM read GB | HBM write GB | HBM GB/s | of 8 TB/s | VALU / wave time | LDS / wave time | issuing / wave time | waiting / wave time | LDS bank conflict / LDS active | VALU insts per input byte |",
     "|---|---|---|---|---|---|---|---|---|---|---|---|"]
for k in kernels:
    if k not in ms1k:
        continue
    c = C.get(k, {})
    rd, wr = F.get(k, 0) * 2048 / 1e9, W.get(k, 0) * 1024 / 1e9
    wave = c.get("SQ_WAVE_CYCLES", 0) or 1
    lds_act = c.get("SQ_LDS_IDX_ACTIVE", 0) or 1
    rate = (rd + wr) / (ms1k[k] * 1e-3)
    T.append(f"| {k} | {ms1k[k]:.3f} | {rd:.2f} | {wr:.2f} | {rate:.0f} | {rate / 8000:.3f} | "
             f"{c.get('SQ_ACTIVE_INST_VALU', 0) / wave:.3f} | {c.get('SQ_ACTIVE_INST_LDS', 0) / wave:.3f} | "
             f"{c.get('SQ_ACTIVE_INST_ANY', 0) / wave:.3f} | {c.get('SQ_WAIT_INST_ANY', 0) / wave:.3f} | "
             f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / lds_act:.3f} | {c.get('SQ_INSTS_VALU', 0) * 64 / N1K:.1f} |")
T.append("\nRaw counter averages per dispatch:\n")
T.append("| kernel | " + " | ".join(sorted({n for v in C.values() for n in v})) + " |")
T.append("|---|" + "---|" * len({n for v in C.values() for n in v}))
for k in kernels:
    if k in C:
        T.append(f"| {k} | " + " | ".join(f"{C[k].get(n, 0):.3g}" for n in sorted({n for v in C.values() for n in v})) + " |")
open(P(f"{tag}_counters.md"), "w").write("\n".join(T) + "\n")

# ---- summary ----
rows = [r for r in csv.DictReader(open(stats_file)) if "m4g" in r["Name"]]
L = [f"# Round {tag[1:]} profile summary (1 x MI355X)\n",
     "Made by `tools/profile_run_r03.sh` + `tools/make_profile_summary_r03.py`.",
     "Trace: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-overlap --no-cpu-baseline --no-end-to-end --verify-chunks 2 --flate-chunks 0 --writer-mib 0`",
     "(`--no-overlap`: the one-stream form of the pipeline, every kernel alone on the GPU -- the durations `stage_ms` and `roofline` of every bench line are made of; the default run overlaps two halves of the batch on two streams, DESIGN.md section 5).",
     f"(4096 x 1 MiB synthetic text chunks, params G; the run also makes the gpu_encode leg, hence k6 and k5_histograms<true>; the kernels of the flate and writer legs, 256 chunks: `profiles/{tag}_kernel_stats_legs.csv`).",
     f"Raw: `profiles/{tag}_kernel_stats.csv`; bench line of the plain `python bench.py` run: `profiles/{tag}_bench.json`;",
     f"counter table: `profiles/{tag}_counters.md`; HBM-side bytes per input byte: `profiles/traffic.json`.\n",
     "| kernel | calls | avg ms | % |\n|---|---|---|---|"]
for r in rows:
    L.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |")
e2e, ge, ce = b.get("end_to_end", {}), b.get("gpu_encode", {}), b.get("cpu_encode", {})
L.append(f"\nbench.py (HIP events on the launch stream): stage_ms = {b['stage_ms']}, value = {b['value']} GB/s "
         f"({b['ms_per_step']} ms per 4 GiB step; one stream: {b.get('overlap', {}).get('one_stream_ms_per_step')} ms; outputs of the two forms equal: "
         f"{b.get('overlap', {}).get('outputs_equal_one_stream_form')}), roofline.achieved = {b['roofline']['achieved']} GB/s of 8000 "
         f"({b['roofline']['frac']}; kernel {b['roofline']['kernel']}), roofline.traffic = {b['roofline']['traffic']} GB, "
         f"whole path {b['roofline'].get('whole_path_frac')}; parity checked on {b['config']['parity_checked_chunks']} of "
         f"{b['config']['parity_checked_of']} chunks; cpu_baseline = {b['cpu_baseline']['value']} GB/s on "
         f"{b['cpu_baseline']['cores']} threads ({b['cpu_baseline'].get('value_1thread')} on one); compression ratio "
         f"{b['config']['compression_ratio']}.")
ws = b["config"].get("workspace")
if ws:
    L.append(f"Plan workspace (m4g_batch_info): {ws['bytes'] / 2**30:.2f} GiB = {ws['bytes_per_input_byte']} bytes per input byte, "
             f"{ws['slot']} Match records per sub-block slot (worst case {ws['slot_worst']}), re-runs after a slot overflow: {ws['reruns']}.")
L.append(f"Encoder: CPU Encode (restated, one thread) {ce.get('s_per_GiB')} s per GiB; m4g_batch_encode over the 4 GiB "
         f"{ge.get('ms_total')} ms ({ge.get('value')} GB/s: host prefix codes {ge.get('ms_host_prefix_codes')} ms, K6 "
         f"{ge.get('ms_k6_emit_bits')} ms), stream {ge.get('stream_bytes')} bytes, ratio {ge.get('compression_ratio')}.")
L.append(f"Host-memory API (PCIe both ways): {e2e.get('value')} GB/s from pageable memory, {e2e.get('value_pinned')} from "
         f"m4g_host_alloc buffers (512 chunks, set-up included), {e2e.get('value_pinned_persistent')} on the persistent handle "
         f"({e2e.get('persistent_sample_chunks')} chunks).\n")
gf, gw = b.get("gpu_flate", {}), b.get("gpu_writer", {})
L.append(f"flate (level 6 M4 fields, {gf.get('chunks')} chunks as one gzip stream): match finding {gf.get('find_matches_value')} GB/s, "
         f"m4g_batch_flate_encode {gf.get('encode_ms_total')} ms = {gf.get('encode_value')} GB/s (host codes {gf.get('encode_ms_host_codes')} ms, "
         f"k6f {gf.get('encode_ms_k6f_tokens')} ms), ratio {gf.get('compression_ratio')}, inflates: {gf.get('gzip_stream_inflates_to_input')}, "
         f"restated CPU encoder {gf.get('cpu_encode_s_per_GiB')} s per GiB.")
L.append(f"m4g_writer (matchfinder.Writer on the GPU, ONE stream of {gw.get('stream_bytes_in')} bytes, 64 KiB blocks, pageable host memory "
         f"in and out): brotli params G {gw.get('brotli_G')}, gzip level 6 {gw.get('gzip_level6')}; more: `profiles/{tag}_writer_probe.log`, "
         f"two-call form: `profiles/{tag}_compress_probe.log`.\n")
gb = b.get("gpu_bargain", {})
L.append(f"Bargain1 (NewWriterV2 levels 5-6; {gb.get('chunks')} streams of 1 MiB in 64 KiB blocks = {gb.get('blocks')} blocks, one wavefront "
         f"each): {gb.get('value')} GB/s ({gb.get('ms')} ms: {gb.get('stage_ms')}), {gb.get('bytes_per_match')} bytes per Match, first "
         f"{gb.get('chunks_compared')} streams equal the restated finder: {gb.get('first_chunks_equal_restated_finder')}; the restated finder "
         f"on one core {gb.get('cpu_restated_MB_s_one_core')} MB/s; Bargain2 / Bargain3 and other sizes: `profiles/{tag}_bargain_probe.log`; NewWriterV2(dst, 5) + Write + Close on one stream, "
         f"host to host: {gb.get('new_writer_v2_level5')}; levels 5..9 "
         f"(m4g_brotli_compress_v2): `profiles/{tag}_v2_probe.log`.\n")
L.append("Other workloads (1024 x 1 MiB chunks, same run):\n")
L.append(f"* low-entropy binary log, params G (BASELINE config 5): {bl['value']} GB/s, stage_ms {bl['stage_ms']}, ratio {bl['config']['compression_ratio']}, roofline.traffic {bl['roofline']['traffic']} GB")
L.append(f"* text, params D (ChainLength 64, HashLen 5; BASELINE config 4): {bd['value']} GB/s, stage_ms {bd['stage_ms']}, ratio {bd['config']['compression_ratio']}, roofline.traffic {bd['roofline']['traffic']} GB")
L.append(f"* inputs that defeat the speculation: `profiles/{tag}_kind_probe.log`; one stream in 64 KiB blocks: `profiles/{tag}_stream_probe.log`; persistent host batch: `profiles/{tag}_hostbatch_probe.log`\n")
tr = json.load(open(P("traffic.json")))
L.append("## HBM-side traffic, bytes per input byte (fetch x2 corrected + write; `profiles/traffic.json`)\n")
L.append("| workload | K1 | K3 | K4 | K5 |\n|---|---|---|---|---|")
for key in ("text/G", "text/D", "log/G"):
    if key in tr:
        e = tr[key]
        L.append(f"| {key} | {e['k1_chain_build']} | {e['k3_parse']} | {e['k4_compact']} | {e['k5_histograms']} |")
L.append("\nAlgorithmic bytes per input byte (SURVEY 8d): 1 + 12 M / N + 4096 / N = about 2.0 for text.")
open(P(f"{tag}_summary.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L[:14]))
