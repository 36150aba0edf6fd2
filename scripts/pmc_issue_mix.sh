#!/bin/bash
# Counter-based view of every forward kernel (two SQ counter passes, --kernel-trace only) -> gpurun_out/pmc_issue_mix.json
# per kernel: launches per forward, VALU / MFMA / LDS / VMEM / SALU instructions per wave and per MFMA, the share of wave cycles spent waiting,
# LDS bank-conflict share and the MATRIX-CORE UTILISATION  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)
# (the counter advances 32 per 32x32x16 MFMA wave-instruction = the cycles one SIMD's matrix core is occupied; kernel cycles =
# GRBM_GUI_ACTIVE, cross-checked against SQ_BUSY_CYCLES / 32 shader engines).  The stack-wide figure (all conv kernels of one forward) is what
# bench.py stamps into roofline.mfma_busy_frac while csrc/ + engine.py still hash to `kernel_src_sha16`.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp Y5_TUNE_CACHE=/tmp/tc_fwd.json Y5_GRAPH=0
# the timed bench run's tile choices and in-situ decisions (scripts/r6_final.sh exports the file): the profiled plan is the timed plan
[ -n "$Y5_SEED_TUNE_CACHE" ] && [ -f "$Y5_SEED_TUNE_CACHE" ] && cp "$Y5_SEED_TUNE_CACHE" "$Y5_TUNE_CACHE"
# fusion decisions forced on (what the timing at plan build picks on this part): the profiled process launches no fused-vs-unfused timing kernels
export Y5_FUSED_K3PW=1 Y5_FUSED_CV3=1 Y5_FUSED_HEAD=1 Y5_FUSED_FRONT=1
N=5
python scripts/forward_only.py 2 > /dev/null 2>&1
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcm_$i
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OLDPWD/gpurun_out/pmcm_$i" -o p -- python "$OLDPWD/scripts/forward_only.py" $N > "$OLDPWD/gpurun_out/pmcm_$i.log" 2>&1)
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, '.')
N = $N
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
for f in glob.glob('gpurun_out/pmcm_*/**/*counter_collection.csv', recursive=True):
    first = 'pmcm_1' in f
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'y5_' not in k: continue
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        if first and r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id']); cnt[k] += 1
            dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3   # us, pass 1 (the pass the busy counters come from)
conv = lambda k: any(s in k for s in ('conv_igemm', 'conv_h3', 'conv_pw', 'conv_k3', 'conv_stem', 'conv_bneck', 'conv_front', 'conv_c3', 'sppf_cv1_pool', 'conv_headk', 'conv_g8'))
NSIMD = 1024.0
MAX_GHZ = 2.4   # data-sheet shader clock of the part: no kernel can have had more cycles than duration x this
def util(d):
    gui = d.get('GRBM_GUI_ACTIVE', 0.0)
    cyc = d.get('SQ_BUSY_CYCLES', 0.0) / 32.0          # SQ_BUSY_CYCLES is summed over the 32 shader engines
    # GRBM_GUI_ACTIVE may come back per XCD (x8): take the reading that agrees with the SQ one
    if gui > 0 and cyc > 0:
        gui = min((gui, gui / 8.0), key=lambda g: abs(g - cyc))
    base = gui if gui > 0 else cyc
    return (d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (NSIMD * base) if base > 0 else None), base
out = []
for k, d in tot.items():
    w = d.get('SQ_WAVES', 0) or 1
    wc = d.get('SQ_WAVE_CYCLES', 0) or 1
    mf = d.get('SQ_INSTS_MFMA', 0)
    u, cyc = util(d)
    #  the cycle denominator is only trusted while it implies a clock the part can run at; otherwise -- and always as a second
    # figure -- utilisation against the cycles the kernel's DURATION allows at the maximum clock (a lower bound on the true busy fraction)
    ghz = cyc / max(dur[k], 1e-9) * 1e-3 if cyc else None
    cyc_time = dur[k] * 1e-6 * MAX_GHZ * 1e9
    u_time = d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (NSIMD * cyc_time) if cyc_time > 0 else None
    plausible = ghz is not None and ghz <= 2.45
    if not plausible:
        u, cyc = u_time, cyc_time
    out.append({"kernel": k[:140], "conv": conv(k), "launches_per_forward": round(cnt[k] / N, 1), "us_per_forward_under_pmc": round(dur[k] / N, 1),
                "waves_per_launch": round(w / max(cnt[k], 1)), "mfma_busy_frac": round(u, 4) if u is not None else None,
                "mfma_busy_frac_vs_duration_at_max_clock": round(u_time, 4) if u_time is not None else None,
                "effective_clock_ghz": round(ghz, 3) if ghz else None, "cycle_counter_plausible": plausible,
                "lds_busy_frac": round(d.get('SQ_LDS_IDX_ACTIVE', 0.0) / (256.0 * cyc), 4) if cyc else None,
                "valu_per_wave": round(d.get('SQ_INSTS_VALU', 0) / w), "mfma_per_wave": round(mf / w), "lds_per_wave": round(d.get('SQ_INSTS_LDS', 0) / w),
                "vmem_per_wave": round(d.get('SQ_INSTS_VMEM', 0) / w), "salu_per_wave": round(d.get('SQ_INSTS_SALU', 0) / w),
                "valu_per_mfma": round(d.get('SQ_INSTS_VALU', 0) / mf, 2) if mf else None, "salu_per_mfma": round(d.get('SQ_INSTS_SALU', 0) / mf, 2) if mf else None,
                "lds_per_mfma": round(d.get('SQ_INSTS_LDS', 0) / mf, 2) if mf else None,
                "active_valu_over_wave_cycles": round(d.get('SQ_ACTIVE_INST_VALU', 0) / wc, 3),
                "wait_inst_over_wave_cycles": round(d.get('SQ_WAIT_INST_ANY', 0) / wc, 3),
                "lds_bank_conflict_over_lds_active": round(d.get('SQ_LDS_BANK_CONFLICT', 0) / d['SQ_LDS_IDX_ACTIVE'], 3) if d.get('SQ_LDS_IDX_ACTIVE') else None})
out.sort(key=lambda r: -r["us_per_forward_under_pmc"])
ck = [k for k in tot if conv(k)]
busy = sum(tot[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for k in ck)
def kcycles(k):   # per kernel: the counter's cycles where they imply a possible clock, the duration at the maximum clock otherwise
    c = util(tot[k])[1]
    return c if c and c / max(dur[k], 1e-9) * 1e-3 <= 2.45 else dur[k] * 1e-6 * MAX_GHZ * 1e9
cycles = sum(kcycles(k) for k in ck)
import bench
res = {"forwards": N, "kernel_src_sha16": bench.kernel_src_hash(),
       "stack": {"mfma_busy_frac": round(busy / (NSIMD * cycles), 4) if cycles else None,
                 "mfma_busy_frac_vs_duration_at_max_clock": round(busy / (NSIMD * sum(dur[k] for k in ck) * 1e-6 * MAX_GHZ * 1e9), 4) if ck else None, "conv_us_per_forward_under_pmc": round(sum(dur[k] for k in ck) / N, 1),
                 "mfma_busy_cycles_per_forward": busy / N, "kernel_cycles_per_forward": cycles / N,
                 "definition": "sum over the conv launches of one forward of SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x sum of their kernel cycles); profiled passes "
                               "serialise the launches, so this is the utilisation INSIDE the kernels (no launch gaps)"},
       "kernels": out}
res.update(json.load(open('gpurun_out/forward_only_plan.json')))   # the plan the counters belong to (bench.py compares its kernel families x launches)
json.dump(res, open('gpurun_out/pmc_issue_mix.json', 'w'), indent=1)
print(json.dumps(res["stack"]))
assert all(r["effective_clock_ghz"] is None or r["effective_clock_ghz"] <= 2.45 or not r["cycle_counter_plausible"] for r in out)
for r in out[:16]: print({k: r[k] for k in ("kernel", "launches_per_forward", "us_per_forward_under_pmc", "mfma_busy_frac", "mfma_busy_frac_vs_duration_at_max_clock", "lds_busy_frac", "effective_clock_ghz", "valu_per_mfma", "salu_per_mfma", "lds_per_mfma", "wait_inst_over_wave_cycles", "lds_bank_conflict_over_lds_active")})
PY
find gpurun_out/pmcm_* -name "*.csv" -size +5M -delete
