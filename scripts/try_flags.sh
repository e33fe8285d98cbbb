# usage: bash scripts/try_flags.sh "<flags A>" "<flags B>" ...   (A/B of compile-time variants on the GPU box)
set -e
cd $GRAFT_REPO_ROOT
for F in "$@"; do
  for src in drx_cdae drx_shard drx_generic; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $F -I include -I drecpy_amd/csrc -c drecpy_amd/csrc/$src.hip -o drecpy_amd/csrc/build/$src.hip.o
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o drecpy_amd/libdrx.so drecpy_amd/csrc/build/*.o
  echo "FLAGS=$F"
  python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['phases_ms'].items()})"
done
