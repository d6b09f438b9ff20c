# parity attribution of the golden 481x849 clip (tools/parity_attribution.py), one process per switch
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02_a}
timeout 900 python tools/parity_attribution.py --tag base --out gpurun_out/${TAG}_parity_base.json > gpurun_out/${TAG}_parity_base.log 2>&1; tail -1 gpurun_out/${TAG}_parity_base.log | cut -c1-900
RMEM_P16=1 timeout 600 python tools/parity_attribution.py --tag p16 --rows product,cpuenc,lsttonly --out gpurun_out/${TAG}_parity_p16.json > gpurun_out/${TAG}_parity_p16.log 2>&1; tail -1 gpurun_out/${TAG}_parity_p16.log | cut -c1-600
RMEM_FOLD_BN=0 timeout 600 python tools/parity_attribution.py --tag nofold --rows product --out gpurun_out/${TAG}_parity_nofold.json > gpurun_out/${TAG}_parity_nofold.log 2>&1; tail -1 gpurun_out/${TAG}_parity_nofold.log | cut -c1-400
MIOPEN_DEBUG_CONV_WINOGRAD=0 timeout 600 python tools/parity_attribution.py --tag nowino --rows product,deconly,oracle --out gpurun_out/${TAG}_parity_nowino.json > gpurun_out/${TAG}_parity_nowino.log 2>&1; tail -1 gpurun_out/${TAG}_parity_nowino.log | cut -c1-600
MIOPEN_DEBUG_CONV_WINOGRAD=0 RMEM_FOLD_BN=0 timeout 600 python tools/parity_attribution.py --tag nowino_nofold --rows product --out gpurun_out/${TAG}_parity_nowino_nofold.json > gpurun_out/${TAG}_parity_nowino_nofold.log 2>&1; tail -1 gpurun_out/${TAG}_parity_nowino_nofold.log | cut -c1-400
