"""The canary of the N x N driver's device-to-device hand-over (commet_amd.matrix): a fresh process — nothing of the rank's
state in it, one ROCm runtime, no torch — that imports the first real set a rank is about to import
(commet_readset_import: HIP IPC handles of the owner's device buffers) and leaves.  Exit code 0: the set came across and
holds reads.  The rank waits for this process with a deadline and kills it when it does not answer; it never runs the import
itself first.  Round 3 saw that call hang for good on sets of 50 M reads (never on small ones, which is why the probe set of
the ranks is not enough) in processes that held torch's own ROCm runtime beside the system's.

  python commet_amd/ipc_canary.py <device> <k> <t> <scratch dir> <set numbers, comma separated>
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # started by path from any working directory


def main(argv):
    device, k, t, scratch = int(argv[0]), int(argv[1]), int(argv[2]), argv[3]
    cands = [os.path.join(scratch, f"set{s}.ipc") for s in argv[4].split(",") if s != ""]
    assert "torch" not in sys.modules
    import commet_amd
    deadline = time.monotonic() + float(os.environ.get("COMMET_DIST_TIMEOUT_S", "600"))
    with commet_amd.Context(k=k, t=t, device=device) as ctx:
        while True:
            path = next((p for p in cands if os.path.exists(p)), None)
            if path is not None:
                break
            if time.monotonic() > deadline or not os.path.isdir(scratch):
                return 3
            time.sleep(0.002)
        with open(path, "rb") as fh:
            blob = fh.read()
        rs = commet_amd.ReadSet.import_(ctx, blob)
        ok = rs.num_reads > 0 and sum(rs.file_reads()) == rs.num_reads
        rs.close()
    return 0 if ok else 2


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
