cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call m: which of {1, 8 ranks} x {cold, warm MIOpen user state} changes the clip hashes of bench.py --config clips64?
O=gpurun_out/r06m; mkdir -p $O
run() { # $1 ranks, $2 clips per rank, $3 tag, rest: env
  local n=$1 per=$2 tag=$3; shift 3
  env RMEM_DEVICE_OVERRIDE=0 RMEM_DIST_BACKEND=gloo "$@" timeout 900 python bench.py --gpus $n --config clips64 --clips-per-rank $per --clip-frames 4 2>/dev/null \
    | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', [h[:8] for h in d['clip_sha256']])"
}
C1=$(mktemp -d); C2=$(mktemp -d); C3=$(mktemp -d)
{
run 1 8 "1 rank, cold A       " MIOPEN_USER_DB_PATH=$C1/db MIOPEN_CUSTOM_CACHE_DIR=$C1/cache
run 1 8 "1 rank, same dirs    " MIOPEN_USER_DB_PATH=$C1/db MIOPEN_CUSTOM_CACHE_DIR=$C1/cache
run 8 1 "8 ranks, those dirs  " MIOPEN_USER_DB_PATH=$C1/db MIOPEN_CUSTOM_CACHE_DIR=$C1/cache
run 8 1 "8 ranks, cold B      " MIOPEN_USER_DB_PATH=$C2/db MIOPEN_CUSTOM_CACHE_DIR=$C2/cache
run 8 1 "8 ranks, same dirs B " MIOPEN_USER_DB_PATH=$C2/db MIOPEN_CUSTOM_CACHE_DIR=$C2/cache
run 1 8 "1 rank, dirs B       " MIOPEN_USER_DB_PATH=$C2/db MIOPEN_CUSTOM_CACHE_DIR=$C2/cache
run 1 8 "1 rank, default dirs "
run 8 1 "8 ranks, default dirs"
run 1 8 "1 rank, cold C, FIND_MODE=2 ENFORCE=1" MIOPEN_USER_DB_PATH=$C3/db MIOPEN_CUSTOM_CACHE_DIR=$C3/cache MIOPEN_FIND_MODE=2 MIOPEN_FIND_ENFORCE=1
ls -la $C1/db $C2/db 2>/dev/null | head -20
} 2>&1 | tee $O/world_hash_matrix.txt
