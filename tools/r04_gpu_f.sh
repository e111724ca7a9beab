cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_f
echo "== tiny q4kv2"; timeout 120 tools/bin/dbg_layer_parts 4 2 256 512 22
echo "== tiny q7kv1"; timeout 120 tools/bin/dbg_layer_parts 7 1 256 512 62
echo "== full"; timeout 120 tools/bin/dbg_layer_parts 28 4 3584 18944 3581
