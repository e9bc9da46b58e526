cd $GRAFT_REPO_ROOT/tools/microbench && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/store_stream store_stream.hip 2>&1 | grep -E "error"
/tmp/store_stream
