"""profiles/r<N>_summary.txt: ONE table per mode with, for every shipped kernel symbol, launches and average duration in the bench
(HIP events / rocprofv3 stats), achieved TFLOP/s where the bench knows the FLOPs, and the PMC figures of the same command
(effective clock, MFMA-busy fraction, fabric bytes per launch and TB/s).  usage: python tools/make_summary.py [round, default 3] > profiles/r3_summary.txt"""
import json, os, re, sys
RN = sys.argv[1] if len(sys.argv) > 1 else "4"
R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
def last_json(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
def stats(path):
    rows = {}
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)$", line)
        if m:
            name = m.group(1).replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            rows[name] = (int(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5)))
    return rows
def table(title, stats_file, traffic_file, line):
    print(title)
    st = stats(os.path.join(R, stats_file))
    tr = json.load(open(os.path.join(R, traffic_file))) if os.path.exists(os.path.join(R, traffic_file)) else {}
    tf = {}
    for k in (line or {}).get("kernels", []):
        sym = k["kernel"].split(" M=")[0].split(" B=")[0]
        e = tf.setdefault(sym, [0.0, 0.0])
        e[0] += k["tflops"] * k["total_ms"]; e[1] += k["total_ms"]
    print(f"{'kernel symbol':44s} {'calls':>6s} {'avg us':>8s} {'% time':>6s} {'TFLOP/s':>8s} {'x2500':>6s} {'GHz':>5s} {'MFMA busy':>9s} {'fabric MB':>10s} {'TB/s':>5s}")
    for name, (calls, tot, avg, pct) in list(st.items())[:16]:
        t = tr.get(name) or {}
        tfl = tf.get(name)
        tfs = tfl[0] / tfl[1] if tfl and tfl[1] else None
        mb = t.get("total_bytes")
        us = t.get("avg_us_under_pmc")
        f = lambda v, w, p_: (f"{v:{w}.{p_}f}" if v is not None else " " * (w - 1) + "-")
        print(f"{name[:44]:44s} {calls:6d} {avg:8.1f} {pct:6.2f} {f(tfs, 8, 0)} {f(tfs / 2500 if tfs else None, 6, 3)} "
              f"{f(t.get('clock_ghz'), 5, 2)} {f(t.get('mfma_busy'), 9, 2)} {f(mb / 1e6 if mb else None, 10, 1)} "
              f"{f(mb / us / 1e6 if mb and us else None, 5, 2)}")
    print()
print(f"Round-{RN} per-kernel summary of the SHIPPED symbols (MI355X, CogVideoX-2B 320x480x17f, B=4).  calls / avg us / % time: rocprofv3\n"
      "--kernel-trace --stats of the bench command; TFLOP/s: algorithmic FLOPs / HIP-event time inside bench.py (same command, no\n"
      "profiler); GHz = GRBM_GUI_ACTIVE / 8 / duration, MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GUI_ACTIVE / 8), fabric MB =\n"
      "FETCH_SIZE x 2 + WRITE_SIZE per launch (L2 misses: Infinity Cache + HBM), from separate --pmc passes (tools/pmc_bench.sh).\n")
d = last_json(os.path.join(R, f"r{RN}_bench_line_default.json"))
print(f"headline: {d['ms_per_step']} ms/step, {d['value']} clip-steps/s, {d['achieved_tflops_attn_ffn']} TFLOP/s attention+FFN = {d['frac_mfma_peak_attn_ffn']} x peak\n")
table("== denoise (python bench.py --no-vae) ==", f"r{RN}_bench_novae_kernel_stats_summary.txt", "hbm_traffic.json", d)
t = last_json(os.path.join(R, f"r{RN}_train_2b_line.json"))
print(f"train: {t['ms_per_step']} ms/step, {t['value']} clips/s, peak {t['peak_hbm_gib']} GiB\n")
table("== SFT step (python bench.py --mode train) ==", f"r{RN}_train_kernel_stats_summary.txt", "hbm_traffic_train.json", None)
b1p = os.path.join(R, f"r{RN}_bench_b1_kernel_stats_summary.txt")
if os.path.exists(b1p):
    b1 = last_json(os.path.join(R, f"r{RN}_bench_line_b1.json"))
    print(f"B = 1: {b1['ms_per_step']} ms/step, {b1['achieved_tflops_attn_ffn']} TFLOP/s attention+FFN = {b1['frac_mfma_peak_attn_ffn']} x peak\n")
    table("== denoise, B = 1 (python bench.py --batch 1 --no-vae) ==", f"r{RN}_bench_b1_kernel_stats_summary.txt", "-", b1)
vp = os.path.join(R, f"r{RN}_vae_kernel_stats_summary.txt")
if os.path.exists(vp):
    v = (d.get("vae_decode") or {})
    print(f"VAE decode (tiled, as the reference configures it): {v.get('ms_per_clip')} ms per clip ({v.get('ms_per_clip_untiled')} untiled); 50-step pipeline {v.get('frames_per_sec_50_steps_incl_decode')} frames/s with the decode, {v.get('frames_per_sec_50_steps_excl_decode')} without\n")
    table("== VAE decode (tools/profile_vae.sh) ==", f"r{RN}_vae_kernel_stats_summary.txt", "-", None)
