# L1/L2 behaviour of the long-term P.V kernel (separate --pmc passes, kernel-trace only)
export TMPDIR=/tmp; mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z_0-9]*sum\|TCP_[A-Z_0-9]*sum" | sort -u | tr '\n' ' ' | cut -c1-1500; echo
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_cache_$i -o p -- python tools/kbench.py --only pv --iters 5 > gpurun_out/pmc_cache_$i.log 2>&1
  python - <<PY
import csv, collections, glob
for f in glob.glob("gpurun_out/pmc_cache_$i/*counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pv_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, "mean %.4e" % (sum(v)/len(v)), "n", len(v))
PY
done
