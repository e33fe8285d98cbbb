# A/B: old (pre policy-refactor) drx_cdae.hip vs current, same process conditions
set -e
cd $GRAFT_REPO_ROOT
cp drecpy_amd/libdrx.so /tmp/libdrx_new.so
echo "NEW"; python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['phases_ms'].items()})"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -I drecpy_amd/csrc -c scripts/ab/old_drx_cdae.hip -o /tmp/old_cdae.o
cp drecpy_amd/csrc/build/drx_cdae.hip.o /tmp/new_cdae.o
cp /tmp/old_cdae.o drecpy_amd/csrc/build/drx_cdae.hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o drecpy_amd/libdrx.so drecpy_amd/csrc/build/*.o
echo "OLD"; python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['phases_ms'].items()})"
echo "NEW again"; cp /tmp/libdrx_new.so drecpy_amd/libdrx.so; python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['phases_ms'].items()})"
