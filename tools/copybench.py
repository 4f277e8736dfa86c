import sys
sys.path.insert(0, "/root/repo")
import commet_amd
with commet_amd.Context(k=10) as ctx:
    for mode, name in ((4, "copy"), (5, "fill")):
        for gib in (1, 4, 8):
            ms = ctx.membench(mode, gib << 30, 0)
            moved = (2 if mode == 4 else 1) * (gib << 30)
            print(name, gib, "GiB", round(ms, 3), "ms", round(moved / ms / 1e9, 2), "TB/s (read+write)")
