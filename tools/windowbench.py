"""Windowed random gathers (commet_membench modes >= 100): the rate the tiled search's probe pass can hope for."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import commet_amd
ctx = commet_amd.Context(k=20, t=2)
N = 1 << 31
for table in (512 << 20, 1 << 30):
    for logw in (19, 20, 21, 22, 23):
        for xcd in (0, 1):
            ms = ctx.membench(100 + logw + (1000 if xcd else 0), table, N)
            print(json.dumps({"table_MiB": table >> 20, "window_KiB": (1 << logw) >> 10, "xcd_aware": xcd, "ms": round(ms, 2),
                              "G_gathers_per_s": round(N / ms / 1e6, 1)}), flush=True)
print(json.dumps({"random_whole_table_G_per_s": round(N / ctx.membench(0, 1 << 30, N) / 1e6, 1)}))
ctx.close()
