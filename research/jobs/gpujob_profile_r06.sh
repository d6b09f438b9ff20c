#!/bin/bash
# round-6 profiles (same recipe as rounds 3-4, rocpd or csv output): kernel trace + stats of the default bench command, then PMC passes (separate runs, kernel-trace only,
# as MI355X_MICROARCH.md prescribes) for the fused memory-read kernels; the same counters for 720p K=8 and for 8 clips
# per launch; the other benches of the round.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r06}
O=gpurun_out/$TAG; mkdir -p $O
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F16" "FETCH_SIZE" "WRITE_SIZE")
pmc() {  # name, command...
  local name=$1; shift
  local i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${name}_$i -o p -- "$@" > $O/pmc_${name}_$i.log 2>&1
  done
  NAME=$name O=$O python - <<'PY'
import csv, glob, json, collections, os
name, O = os.environ["NAME"], os.environ["O"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f"{O}/pmc_{name}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if k.startswith("read64") or "read_combine" in k or "mha_" in k or "linear_stream" in k or "dwconv5x5" in k or "layernorm_red2" in k or "LnRed2" in k or "Combine2" in k or "gn2_" in k or "Gn2" in k:
            kk = ("dwconv5x5_split_kernel" if "dwconv5x5" in k else k.split("(")[0].split("<")[0][:48])
            if kk.startswith("void "):
                kk = kk[5:]
            agg[kk][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"note": "rocprofv3 --kernel-trace --pmc <set> in separate passes (4 passes: two SQ sets, FETCH_SIZE, WRITE_SIZE) of the command below; "
               "mean per dispatch over every dispatch of the kernel in the run (pre-roll included); FETCH_SIZE / WRITE_SIZE in KB as reported; "
               "hbm_bytes_per_launch = 2*FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE; SQ_* cycle "
               "counters are quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES (cycles: 32 per 32x32x16 MFMA, 16 per 16x16x32)"}
for kk, cs in agg.items():
    d = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()}
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE"]["mean"] + d["WRITE_SIZE"]["mean"]) * 1024
    if "SQ_WAIT_INST_ANY" in d and "SQ_WAVE_CYCLES" in d:
        d["wait_inst_any_over_wave_cycles"] = d["SQ_WAIT_INST_ANY"]["mean"] / d["SQ_WAVE_CYCLES"]["mean"]
    res[kk] = d
json.dump(res, open(f"{O}/{os.path.basename(O)}_pmc_{name}.json", "w"), indent=1)
print(name, {k: {c: (round(v["mean"] / 1e6, 2) if isinstance(v, dict) else round(v, 3)) for c, v in d.items()} for k, d in res.items() if k != "note"})
PY
  find $O -name "*.csv" -size +1M -delete
}
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# (un-profiled benches FIRST: MIOpen picks slower solvers for the rest of a box's life once rocprofv3 has run on it)
# 3. the benches of the round (un-profiled)
timeout 600 python bench.py > $O/${TAG}_bench_x3.json 2> $O/bench_x3.err
timeout 1500 python bench.py --config 720p_k8 --gap 2 --no-dropin > $O/${TAG}_bench_720p_k8.json 2> $O/bench_720.err
timeout 600 python bench.py --batched --clips-per-gpu 8 --no-cpu-baseline > $O/${TAG}_bench_batched8.json 2> $O/bench_b8.err
timeout 600 python bench.py --clips-per-gpu 2 --no-cpu-baseline --no-dropin > $O/${TAG}_bench_2clips.json 2> $O/bench_2c.err
timeout 900 python bench.py --model r50_aotl --no-dropin > $O/${TAG}_bench_aot.json 2> $O/bench_aot.err
timeout 900 python bench.py --config clips64 > $O/${TAG}_bench_clips64.json 2> $O/bench_c64.err
timeout 900 python bench.py --config clips64 --batched > $O/${TAG}_bench_clips64_batched.json 2> $O/bench_c64b.err
RMEM_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/${TAG}_bench_x3_rccl_world1.json 2> $O/bench_rccl1.err
timeout 300 python tools/lstt_trace.py > $O/${TAG}_lstt_isolated.json 2> $O/lstt_iso.err
timeout 600 python bench.py --model swinb_aotl --no-dropin --cpu-frames 2 > $O/${TAG}_bench_swin.json 2> $O/bench_swin.err
RMEM_DEVICE_OVERRIDE=0 RMEM_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 50 --no-cpu-baseline --no-dropin > $O/${TAG}_bench_gpus2_one_device.json 2> $O/bench_g2.err
# 1. default bench: kernel trace + stats
CMD="python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-dropin"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $TAG -- $CMD > $O/prof.log 2>&1
python tools/prof_summary.py $O/prof/${TAG}_kernel_trace.csv 30 > $O/${TAG}_bench_x3_kernel_stats.md
head -c 1200 $O/prof.log | tail -c 600
cp $O/prof/${TAG}_kernel_stats.csv $O/${TAG}_bench_x3_kernel_stats.csv 2>/dev/null
head -16 $O/${TAG}_bench_x3_kernel_stats.md
find $O/prof -name "*.csv" -size +1M -delete
# 2. PMC: headline config, 720p K=8, 8 clips per launch, AOT
pmc x3 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-dropin
pmc 720p_k8 python bench.py --config 720p_k8 --gap 2 --steps 6 --warmup 2 --no-cpu-baseline --no-dropin
for f in $O/${TAG}_bench_*.json; do python -c "
import json,sys
try:
    d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value'],1), d.get('roofline',{}).get('frac'), d.get('n_gpus'))
except Exception as e: print('$f', 'ERR', e)
"; done
