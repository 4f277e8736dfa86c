# A/B runs of compile-time constants of the ingest path on the GPU box (first / second set upload times of bench.py)
cd $GRAFT_REPO_ROOT
for d in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $d -o commet_amd/libcommet_hip.so commet_amd/csrc/capi.hip -lz || exit 1
  for rep in 1 2; do
  python3 bench.py --no-matrix --cpu-sample 0 --no-probe-count --no-kernel-times --steps 3 > /tmp/o.json 2>/tmp/o.err || { tail -5 /tmp/o.err; exit 1; }
  python3 -c "
import json,sys;b=json.load(open('/tmp/o.json'));d=b['detail'];print(sys.argv[1:], 'first+second', d['upload_and_pack_s'], 'second', d['upload_second_set_s'], 'cold', d['end_to_end_reads_per_s_incl_pcie'], 'warm', d['end_to_end_reads_per_s_incl_pcie_warm_staging'], b['ms_per_step'])" "$d"
  done
done
