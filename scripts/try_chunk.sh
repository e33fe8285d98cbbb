set -e
cd $GRAFT_REPO_ROOT
for C in 32 64 128; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DDRX_CHUNK=$C -I include -I drecpy_amd/csrc -c drecpy_amd/csrc/drx_cdae.hip -o drecpy_amd/csrc/build/drx_cdae.hip.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DDRX_CHUNK=$C -I include -I drecpy_amd/csrc -c drecpy_amd/csrc/drx_shard.hip -o drecpy_amd/csrc/build/drx_shard.hip.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o drecpy_amd/libdrx.so drecpy_amd/csrc/build/*.o
  echo "CHUNK=$C"
  python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phases_ms'])"
done
