#!/bin/bash
# HBM traffic of one forward from the memory-side L2 counters (MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE in
# separate passes, --kernel-trace only; FETCH_SIZE doubled for 16 B/lane streaming reads on gfx950).  -> gpurun_out/pmc_forward.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp Y5_TUNE_CACHE=/tmp/tc_fwd.json Y5_GRAPH=0
# the timed bench run's tile choices and in-situ decisions (scripts/r6_final.sh exports the file): the profiled plan is the timed plan
[ -n "$Y5_SEED_TUNE_CACHE" ] && [ -f "$Y5_SEED_TUNE_CACHE" ] && cp "$Y5_SEED_TUNE_CACHE" "$Y5_TUNE_CACHE"
# fusion decisions forced on (what the timing at plan build picks on this part): the profiled process launches no fused-vs-unfused timing kernels
export Y5_FUSED_K3PW=1 Y5_FUSED_CV3=1 Y5_FUSED_HEAD=1 Y5_FUSED_FRONT=1
N=10
python scripts/forward_only.py 2 > /dev/null 2>&1   # fills the tile-choice cache: the profiled runs launch no timing kernels
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcf_$C
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OLDPWD/gpurun_out/pmcf_$C" -o p -- python "$OLDPWD/scripts/forward_only.py" $N > "$OLDPWD/gpurun_out/pmcf_$C.log" 2>&1)
  echo "pass $C rc=$?"
done
python - <<PY
import csv, glob, collections, json
N = $N
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmcf_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'y5_' not in k: continue
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'FETCH_SIZE': cnt[k] += 1
conv = lambda k: any(s in k for s in ('conv_igemm', 'conv_h3', 'conv_pw', 'conv_k3', 'conv_stem', 'conv_bneck', 'conv_front', 'sppf_cv1_pool', 'conv_headk', 'conv_g8'))
def gb(keys, name, mult): return sum(tot[k][name] for k in keys) * 1024 * mult / N / 1e9   # counters are in KiB
ck = [k for k in tot if conv(k)]; ok = [k for k in tot if not conv(k)]
out = {"forwards": N, "conv_launches_per_forward": sum(cnt[k] for k in ck) / N,
       "conv_fetch_gb_per_forward_x2_corrected": gb(ck, 'FETCH_SIZE', 2), "conv_write_gb_per_forward": gb(ck, 'WRITE_SIZE', 1),
       "other_fetch_gb_per_forward_x2_corrected": gb(ok, 'FETCH_SIZE', 2), "other_write_gb_per_forward": gb(ok, 'WRITE_SIZE', 1)}
out["conv_traffic_gb_per_forward"] = out["conv_fetch_gb_per_forward_x2_corrected"] + out["conv_write_gb_per_forward"]
import sys; sys.path.insert(0, '.')
import bench
out["kernel_src_sha16"] = bench.kernel_src_hash()   # bench.py reports this traffic only while csrc/ + engine.py still hash to it
out.update(json.load(open('gpurun_out/forward_only_plan.json')))   # ... and only for a timed plan of the same kernel families x launches
json.dump(out, open('gpurun_out/pmc_forward.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
find gpurun_out/pmcf_* -name "*.csv" -size +5M -delete
