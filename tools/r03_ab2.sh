# scatter2 (packed) ablations on skewed data + same-address LDS atomics
R=$GRAFT_REPO_ROOT
cd $R
python3 - <<'P'
import commet_amd, json
n = 1 << 30
with commet_amd.Context(k=10) as ctx:
    for mode in (0, 1):
        for words in (1, 2, 8, 128):
            ms = ctx.ldsbench(mode, words, n)
            print("lds", mode, words, round(ms, 3), "ms", round(n / ms / 1e6, 1), "Gops/s", flush=True)
P
for a in 128 256 32 64 416; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DCOMMET_ABLATE=$a -o commet_amd/libcommet_hip.so commet_amd/csrc/capi.hip -lz || exit 1
  (cd /tmp && python3 $R/bench.py --skew 0.1 --no-matrix --cpu-sample 0 --no-probe-count --steps 2 --kt-steps 2 > /tmp/o.json 2>/tmp/o.err) || { tail -5 /tmp/o.err; exit 1; }
  python3 -c "
import json,sys;b=json.load(open('/tmp/o.json'));print('ablate', sys.argv[1:], {k:round(v['ms_per_step'],2) for k,v in b['roofline']['kernels'].items() if k.startswith('part_s')})" "$a"
done
