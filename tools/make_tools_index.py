"""Regenerates tools/README.md: one line per developer script (its first comment / docstring line), current-round scripts first."""
import os, re
T = os.path.dirname(os.path.abspath(__file__))
CUR = {'refresh_r6.sh', 'r6_check1.sh', 'r6_check2.sh', 'r6_d8r192_ab.sh', 'r6_dbg.sh', 'r6_ln_ab.sh', 'r6_lna_abl.sh', 'r6_epi_price.sh', 'r6_rpf_ab.sh', 'check_async_load_hazard.py', 'variants.sh', 'refresh_r5.sh', 'make_summary.py', 'make_tools_index.py', 'profile_bench.sh', 'profile_vae.sh', 'pmc_bench.sh', 'pmc_bench.py', 'multi_rank_check.sh',
       'bench_line_brief.py', 'gpu_suite.sh', 'full_check.sh', 'make_leaf_kat.py', 'make_vae_naive.py', 'kbench_gemm.cpp', 'kbench_attn.cpp', 'model_ab.sh',
       'time_attn_bwd.py', 'bwd_ab.sh', 'bwd_abl.sh', 'bwd_abl_run.sh', 'attn_bwd_seg_trace.py', 'probe_lds_bcast.cpp', 'attn_fwd_ab.sh', 'power_gemm.py',
       'power_gemm.sh', 'power_attn.py', 'd8_ab.sh', 'd8_abl.sh', 'd8_zero.sh', 'd8_variants.sh', 'd8_b1.sh', 'd8_prev_ab.sh', 'd8_rs_ab.sh', 'd8_rlds_ab.sh',
       'w64_ab.sh', 'b1_ab.sh', 'chains_ab.sh', 'r192_ab.sh', 'r192_tiles.py', 'pqkv_ab.sh', 'ln_rows_b1.sh', 'hipblaslt_compare.py', 'lib_ab.sh', 'ln_packed_ab.sh', 'probe_frag_loads.cpp', 'power_abl.sh', 'vae_bench.py', 'train_memory.py'}
def first(p):
    try: L = open(p, errors='ignore').read().splitlines()
    except Exception: return ''
    for l in L[:12]:
        s = l.strip()
        if s.startswith('#!') or not s: continue
        s = re.sub(r'^(#|//|"""|\'\'\')\s*', '', s).rstrip('"\'').strip()
        if s: return s[:170]
    return ''
rows = [(f, f in CUR, first(os.path.join(T, f))) for f in sorted(os.listdir(T)) if os.path.isfile(os.path.join(T, f)) and f != 'README.md']
out = ['# tools/ index', '',
       'Developer scripts: nothing here is imported by the package, the tests or `bench.py`.  Most are one-off A/B or probe scripts kept so that a',
       'number quoted in `DESIGN.md` / `profiles/` can be re-measured; scripts of earlier rounds may need variant builds (`tools/bin/dv_*`,',
       "made by `*_variants.sh`) or macros that only existed in that round's kernel source (the profile file they produced says which).",
       '`python tools/make_tools_index.py` regenerates this file.', '',
       '## Used by the current round (rounds 5-6)', '', '| script | what it does |', '|---|---|']
out += ['| `%s` | %s |' % (f, d.replace('|', '/')) for f, c, d in rows if c]
out += ['', '## Earlier rounds (historical; results in `profiles/r1_*` ... `r4_*`, narrative in `profiles/HISTORY.md`)', '', '| script | what it does |', '|---|---|']
out += ['| `%s` | %s |' % (f, d.replace('|', '/')) for f, c, d in rows if not c]
open(os.path.join(T, 'README.md'), 'w').write('\n'.join(out) + '\n')
