"""Host-side sharding of the N x N comparison over ranks (one process per GPU).

The path has no exchange step: the unit of work is one
index_and_search invocation, and Commet.py's job DAG (Commet.py:186-240, 570-574) is
    J1(ref)          index S_ref, search S_{ref+1..N-1}
    J2(ref, i)       index S_i restricted to <F>_in_<S_ref>.bv, search S_ref      (needs J1(ref))
    J3(ref, i)       index S_ref restricted to <G>_in_<S_i>.bv, search S_i        (needs J2(ref, i); overwrites J1's output)
for ref < i.  J1(ref) is split per search set so that every (ref, i) pair is one
independent chain J1(ref,i) -> J2(ref,i) -> J3(ref,i): N(N-1)/2 chains, no
ordering between chains, only `.bv` bit-vectors flow inside a chain.  Chains
are dealt to ranks; the only cross-rank operations are a barrier and a MAX
reduction of the elapsed time (on the host, over a small TCP store held by rank 0:
no data-path collective, and no torch in the rank processes)."""
import time


def commet_jobs(n_sets):
    """The N^2-1 invocations of Commet.py in its own order: (kind, index set, [search sets], restrict_to or None)."""
    jobs = []
    for ref in range(n_sets - 1):
        jobs.append(("J1", ref, list(range(ref + 1, n_sets)), None))
        for i in range(ref + 1, n_sets):
            jobs.append(("J2", i, [ref], ref))
            jobs.append(("J3", ref, [i], i))
    return jobs


def pair_chains(n_sets):
    """One chain per unordered pair (ref, i), ref < i: [J1(ref,i), J2(ref,i), J3(ref,i)]."""
    chains = []
    for ref in range(n_sets - 1):
        for i in range(ref + 1, n_sets):
            chains.append([("J1", ref, [i], None), ("J2", i, [ref], ref), ("J3", ref, [i], i)])
    return chains


def assign_chains(chains, world_size, rank, cost=None):
    """Longest-processing-time-first dealing of chains to ranks; deterministic on every rank."""
    if cost is None:
        cost = [1.0] * len(chains)
    order = sorted(range(len(chains)), key=lambda c: (-cost[c], c))
    load = [0.0] * world_size
    mine = []
    for c in order:
        r = min(range(world_size), key=lambda x: (load[x], x))
        load[r] += cost[c]
        if r == rank:
            mine.append(c)
    return sorted(mine)


def assign_pairs_contiguous(cost, world_size):
    """Cuts the row-major list of (ref, i) pairs into world_size CONTIGUOUS runs of about equal cost; returns
    [range(lo, hi)] per rank.  Contiguous runs keep a rank's pairs on few reference sets, so that J1's index of
    S_ref is built once per (rank, ref) and a rank touches few sets; boundaries fall where the running cost crosses
    r / world_size of the total (each run is within one pair's cost of the ideal share)."""
    n, total = len(cost), float(sum(cost))
    cuts, acc, c = [0], 0.0, 0
    for r in range(1, world_size):
        target = total * r / world_size
        while c < n and acc + cost[c] / 2.0 <= target:
            acc += cost[c]
            c += 1
        cuts.append(c)
    cuts.append(n)
    return [range(cuts[r], cuts[r + 1]) for r in range(world_size)]


def assign_owners(n_sets, world_size, rank_cost):
    """Which rank parses which set (every set is parsed once on the node): set s by rank s % world_size for the first
    world_size * (n_sets // world_size) sets; the sets left over — the ranks that take them parse one set more than the
    others — go to the ranks whose pairs cost least (rank_cost[r], ties to the lower rank), one each."""
    owner = [s % world_size for s in range(n_sets)]
    full = world_size * (n_sets // world_size)
    if n_sets >= world_size:
        by_slack = sorted(range(world_size), key=lambda r: (rank_cost[r], r))
        for j, s in enumerate(range(full, n_sets)):
            owner[s] = by_slack[j % world_size]
    return owner


# ---- ranks of one job: barrier / gather / MAX / SUM on the host, no torch --------------------------------------
#
# The path has no data-path collective; what the ranks of a job tell each other are a few small host
# objects (who took which pairs, elapsed seconds, a scratch directory's name).  The default backend is a key / value
# store held by rank 0 on MASTER_ADDR : MASTER_PORT + 1 ... (a launcher such as torch.distributed.run keeps MASTER_PORT
# itself for its own store) which the ranks reach over TCP: nothing but the standard library is imported, so a rank
# process holds ONE ROCm runtime — the system's, through libcommet_hip.so — and not a second one from torch's wheel
# beside it (with both in a process commet_readset_import of a large set did not return).
# backend="gloo" (or COMMET_RANKS_BACKEND=gloo) keeps the torch.distributed form.

_MAGIC = b"COMMETRZ1"
_PORT_SPAN = 32


def _rdzv_token():
    import hashlib
    import os
    tok = os.environ.get("COMMET_RDZV_TOKEN") or ":".join(os.environ.get(v, "") for v in (
        "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "WORLD_SIZE"))
    return hashlib.sha256(tok.encode()).digest()


def _send_frame(sock, *parts):
    import struct
    msg = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
    sock.sendall(struct.pack("<IQ", len(parts), len(msg)) + msg)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("rendezvous connection closed")
        buf += chunk
    return bytes(buf)


def _recv_frame(sock):
    import struct
    nparts, total = struct.unpack("<IQ", _recv_exact(sock, 12))
    msg = _recv_exact(sock, total)
    parts, pos = [], 0
    for _ in range(nparts):
        (ln,) = struct.unpack_from("<Q", msg, pos)
        parts.append(msg[pos + 8:pos + 8 + ln])
        pos += 8 + ln
    return parts


class _Store:
    """Rank 0's key / value store: SET, GET (waits for the key), DEL (by prefix), BYE.  Values are opaque bytes.  Every rank
    connects once; a connection that drops without BYE means that rank died: every waiting and later GET is then answered
    with an error, so the other ranks fail within seconds instead of at a timeout."""

    def __init__(self, world, token, host, first_port):
        import socket
        import threading
        self.world, self.token = world, token
        self.kv, self.cv = {}, threading.Condition()
        self.lost = None
        self.seen, self.byes = set(), 0
        self.sock, self.port = None, None
        err = None
        for port in range(first_port, first_port + _PORT_SPAN):
            sk = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                sk.bind((host, port))
                sk.listen(world + 8)
                self.sock, self.port = sk, port
                break
            except OSError as ex:
                err = ex
                sk.close()
        if self.sock is None:
            raise RuntimeError(f"rendezvous: no free port in {first_port}..{first_port + _PORT_SPAN - 1} on {host!r}: {err}")
        self.thread = threading.Thread(target=self._accept, name="commet-rdzv", daemon=True)
        self.thread.start()

    def _accept(self):
        import threading
        while True:
            try:
                conn, _ = self.sock.accept()
            except OSError:
                return                                           # closed
            threading.Thread(target=self._serve, args=(conn,), daemon=True).start()

    def _serve(self, conn):
        import socket
        import struct
        rank, said_bye = None, False
        try:
            conn.settimeout(5.0)
            hello = _recv_exact(conn, len(_MAGIC) + 32 + 4)
            (r,) = struct.unpack("<I", hello[-4:])
            with self.cv:
                ok = (hello[:len(_MAGIC)] == _MAGIC and hello[len(_MAGIC):-4] == self.token and r < self.world
                      and r not in self.seen and self.lost is None)
                if ok:
                    self.seen.add(r)
                    rank = r
            conn.sendall(b"OK" if ok else b"NO")
            if not ok:
                return
            conn.settimeout(None)
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            while True:
                parts = _recv_frame(conn)
                op = parts[0]
                if op == b"SET":
                    with self.cv:
                        self.kv[parts[1]] = parts[2]
                        self.cv.notify_all()
                    _send_frame(conn, b"OK")
                elif op == b"GET":
                    with self.cv:
                        while parts[1] not in self.kv and self.lost is None:
                            self.cv.wait()
                        if parts[1] in self.kv:
                            _send_frame(conn, b"OK", self.kv[parts[1]])
                        else:
                            _send_frame(conn, b"LOST", str(self.lost).encode())
                elif op == b"DEL":
                    with self.cv:
                        for key in [x for x in self.kv if x.startswith(parts[1])]:
                            del self.kv[key]
                    _send_frame(conn, b"OK")
                elif op == b"CHK":                               # is the job still whole?
                    with self.cv:
                        lost = self.lost
                    if lost is None:
                        _send_frame(conn, b"OK")
                    else:
                        _send_frame(conn, b"LOST", str(lost).encode())
                elif op == b"ABORT":                             # a rank gives the job up: every wait ends with an error, now and later
                    with self.cv:
                        if self.lost is None:
                            self.lost = f"{rank} (gave up: {parts[1].decode(errors='replace')})"
                        self.cv.notify_all()
                    _send_frame(conn, b"OK")
                elif op == b"BYE":
                    said_bye = True
                    with self.cv:
                        self.byes += 1
                        self.cv.notify_all()
                    _send_frame(conn, b"OK")
                    return
        except (OSError, ConnectionError, struct.error):
            pass
        finally:
            if rank is not None and not said_bye:
                with self.cv:
                    if self.lost is None:
                        self.lost = rank
                    self.cv.notify_all()
                try:
                    self.sock.close()                            # a job that lost a rank takes no new connection
                except OSError:
                    pass
            try:
                conn.close()
            except OSError:
                pass

    def wait_byes(self, seconds):
        import time
        deadline = time.monotonic() + (seconds if self.lost is None else min(seconds, 3.0))   # (a failed job: the others are told, then out)
        with self.cv:
            while self.byes < self.world and time.monotonic() < deadline:
                self.cv.wait(0.1)
        try:
            self.sock.close()
        except OSError:
            pass


# ---- what the ranks tell each other, on the wire ------------------------------------------------------------------
# Small host objects only (numbers, strings, a 264-byte descriptor, dicts of pair -> count): JSON with three tags for what JSON
# lacks — bytes, tuples, dicts whose keys are not strings.  Nothing that arrives over the socket is ever executed or
# unpickled: a local user who guesses the job token can make a rank fail, not run code in it.
def _to_wire(o):
    if o is None or isinstance(o, (bool, str)):
        return o
    if isinstance(o, int):
        return o
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else {"__f__": repr(o)}
    if isinstance(o, (bytes, bytearray, memoryview)):
        import base64
        return {"__b__": base64.b64encode(bytes(o)).decode("ascii")}
    if isinstance(o, tuple):
        return {"__t__": [_to_wire(x) for x in o]}
    if isinstance(o, (list, set, frozenset)):
        return [_to_wire(x) for x in (sorted(o) if isinstance(o, (set, frozenset)) else o)]
    if isinstance(o, dict):
        if all(isinstance(k, str) and not k.startswith("__") for k in o):
            return {k: _to_wire(v) for k, v in o.items()}
        return {"__d__": [[_to_wire(k), _to_wire(v)] for k, v in o.items()]}
    item = getattr(o, "item", None)              # numpy scalars
    if callable(item) and getattr(o, "shape", None) == ():
        return _to_wire(item())
    tolist = getattr(o, "tolist", None)          # small numpy arrays
    if callable(tolist):
        return _to_wire(tolist())
    raise TypeError(f"rendezvous: cannot send a {type(o).__name__} between ranks (numbers, strings, bytes, lists, tuples, dicts only)")


def _from_wire(o):
    if isinstance(o, list):
        return [_from_wire(x) for x in o]
    if isinstance(o, dict):
        if len(o) == 1:
            (k, v), = o.items()
            if k == "__b__":
                import base64
                return base64.b64decode(v)
            if k == "__t__":
                return tuple(_from_wire(x) for x in v)
            if k == "__d__":
                return {_hashable(_from_wire(a)): _from_wire(b) for a, b in v}
            if k == "__f__":
                return float(v)
        return {k: _from_wire(v) for k, v in o.items()}
    return o


def _hashable(k):
    return tuple(_hashable(x) for x in k) if isinstance(k, list) else k


def _encode_value(obj):
    import json
    return json.dumps(_to_wire(obj), separators=(",", ":"), allow_nan=False).encode("utf-8")


def _decode_value(body):
    import json
    return _from_wire(json.loads(body.decode("utf-8")))


def pick_device(local_rank, n_devices=None):
    """The HIP device of a rank: COMMET_FORCE_DEVICE when set (a debugging aid: several ranks on one GPU; never set by the
    driver), else LOCAL_RANK modulo the devices this process sees — a launcher that hands every rank ONE visible device
    (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank) must end on device 0 of each, not on device LOCAL_RANK, which
    such a process does not have.  n_devices None or 0 (count unknown: nothing has asked the library): LOCAL_RANK as it is."""
    import os
    forced = os.environ.get("COMMET_FORCE_DEVICE")
    if forced is not None:
        return int(forced)
    return int(local_rank) % int(n_devices) if n_devices else int(local_rank)


class Ranks:
    """Barrier / gather / MAX / SUM over the ranks of a job (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT from the
    environment, as torch.distributed.run and bench.py's own launcher set them).  world_size 1 needs nothing.
    backend: "tcp" (default: standard library only, see above) or "gloo" (torch.distributed); None reads
    COMMET_RANKS_BACKEND."""

    def __init__(self, backend=None):
        import os
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = backend or os.environ.get("COMMET_RANKS_BACKEND", "tcp")
        if self.backend not in ("tcp", "gloo"):
            raise ValueError(f"unknown ranks backend {self.backend!r} (tcp or gloo)")
        self.timeout_s = float(os.environ.get("COMMET_DIST_TIMEOUT_S", "600"))   # a rank that dies must not leave the others waiting for long
        self.dist = None
        self.failed = False                              # a rank of the job has given up (or vanished): no collective will complete any more
        self._sock, self._store, self._seq = None, None, 0
        if self.world > 1 and self.backend == "gloo":
            self._init_gloo()
        elif self.world > 1:
            self._init_tcp()

    # -- gloo ----------------------------------------------------------------------------------------------------
    def _init_gloo(self):
        import os
        import torch.distributed as dist
        if not dist.is_initialized():
            import datetime
            tmo = datetime.timedelta(seconds=self.timeout_s)
            # gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected to ..."): keep them out of a
            # caller's result stream (bench.py prints one JSON line there) by lending fd 1 to stderr meanwhile
            import sys
            sys.stdout.flush()
            saved = os.dup(1)
            try:
                os.dup2(2, 1)
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world, timeout=tmo)
                dist.barrier()          # the connections are made (and announced) at the first collective
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
        self.dist = dist

    # -- tcp -----------------------------------------------------------------------------------------------------
    def _init_tcp(self):
        import os
        import socket
        import struct
        import threading
        import time
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        first = int(os.environ.get("MASTER_PORT", "29500")) + 1
        token = _rdzv_token()
        self._lock = threading.Lock()
        if self.rank == 0:
            local = addr in ("127.0.0.1", "localhost", "::1")
            self._store = _Store(self.world, token, "127.0.0.1" if local else "", first)
        deadline = time.monotonic() + self.timeout_s
        hello = _MAGIC + token + struct.pack("<I", self.rank)
        ports = [self._store.port] if self._store else list(range(first, first + _PORT_SPAN))
        host = "127.0.0.1" if self._store else addr
        while self._sock is None:
            for port in ports:
                sk = None
                try:
                    sk = socket.create_connection((host, port), timeout=2.0)
                    sk.sendall(hello)
                    if _recv_exact(sk, 2) == b"OK":
                        sk.settimeout(self.timeout_s)
                        sk.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self._sock = sk
                        break
                    sk.close()
                except (OSError, ConnectionError):
                    if sk is not None:
                        sk.close()
            if self._sock is None:
                if time.monotonic() > deadline:
                    raise RuntimeError(f"rendezvous: rank {self.rank} found no store of this job on {host}:{first}..{first + _PORT_SPAN - 1}")
                time.sleep(0.02)
        self.barrier()

    def _call(self, *parts):
        with self._lock:
            try:
                _send_frame(self._sock, *parts)
                rep = _recv_frame(self._sock)
            except (OSError, ConnectionError) as ex:
                self.failed = True
                raise RuntimeError(f"rendezvous: rank {self.rank} lost the store ({type(ex).__name__}: {ex}); a peer has died or "
                                   f"nothing arrived within {self.timeout_s:.0f} s") from None
        if rep[0] == b"LOST":
            self.failed = True
            who = rep[1].decode()
            raise RuntimeError(f"rendezvous: rank {who}" + ("" if "gave up" in who else " left the job without saying goodbye"))
        return rep

    def check(self):
        """Raises if a rank of the job has given up or vanished (for waits that do not go through the store: files another rank
        is to publish).  Cheap; any thread."""
        if self._sock is not None:
            self._call(b"CHK")

    def abort(self, reason):
        """This rank gives the job up (it raised): the other ranks' waits, now and later, end with an error naming it — nobody is left
        in a barrier until a timeout.  gloo has no such thing: there a failing rank leaves the process (matrix.main) and the launcher
        ends the group."""
        self.failed = True
        if self._sock is not None:
            try:
                self._call(b"ABORT", str(reason)[:500].encode())
            except RuntimeError:
                pass

    def _dumps(self, obj):
        import hashlib
        import hmac
        body = _encode_value(obj)
        return hmac.new(_rdzv_token(), body, hashlib.sha256).digest() + body

    def _loads(self, blob):
        import hashlib
        import hmac
        if not hmac.compare_digest(blob[:32], hmac.new(_rdzv_token(), blob[32:], hashlib.sha256).digest()):
            raise RuntimeError("rendezvous: a value in the store was not written by a rank of this job")
        return _decode_value(blob[32:])

    def _gather_tcp(self, obj):
        seq = self._seq
        self._seq += 1
        self._call(b"SET", b"g%d/%d" % (seq, self.rank), self._dumps(obj))
        out = [self._loads(self._call(b"GET", b"g%d/%d" % (seq, r))[1]) for r in range(self.world)]
        # whoever is in round `seq` proves that every rank has left round seq - 1 (it has written its value of this round),
        # i.e. has read everything of round seq - 1... but may still be reading it: round seq - 2 is safe to drop
        if self.rank == 0 and seq >= 2:
            self._call(b"DEL", b"g%d/" % (seq - 2))
        return out

    # -- the operations --------------------------------------------------------------------------------------------
    def gather_objects(self, obj):
        if self.world == 1:
            return [obj]
        if self.dist is not None:
            out = [None] * self.world
            self.dist.all_gather_object(out, obj)
            return out
        return self._gather_tcp(obj)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        elif self.world > 1:
            self._gather_tcp(None)

    def max_seconds(self, seconds):
        if self.world == 1:
            return float(seconds)
        if self.dist is not None:
            import torch
            t = torch.tensor([float(seconds)], dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return float(t[0])
        return max(float(x) for x in self._gather_tcp(float(seconds)))

    def sum_int(self, v):
        if self.world == 1:
            return int(v)
        if self.dist is not None:
            import torch
            t = torch.tensor([int(v)], dtype=torch.int64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            return int(t[0])
        return sum(int(x) for x in self._gather_tcp(int(v)))

    def broadcast_object(self, obj, src=0):
        if self.world == 1:
            return obj
        if self.dist is not None:
            box = [obj if self.rank == src else None]
            self.dist.broadcast_object_list(box, src=src)
            return box[0]
        return self._gather_tcp(obj if self.rank == src else None)[src]

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
        elif self._sock is not None:
            try:
                if not self.failed:
                    self.barrier()
                self._call(b"BYE")
            except RuntimeError:
                pass
            self._sock.close()
            self._sock = None
            if self._store is not None:                          # rank 0 stays until every rank has said goodbye
                self._store.wait_byes(30.0)
                self._store = None


def spawn_ranks(n, argv, env=None):
    """Starts the n ranks of a job on this node as plain child processes of a parent that never touches the GPU — RANK,
    LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT and a job token in their environment — and waits
    for them.  A rank that ends non-zero ends the job: the others are terminated (by their process ids) and its code is
    returned; otherwise 0.  The children inherit stdout / stderr."""
    import os
    import secrets
    import socket
    import subprocess
    import time
    with socket.socket() as sk:                      # a free port (the store takes the ones after it)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), COMMET_RDZV_TOKEN=secrets.token_hex(16),
                LOCAL_WORLD_SIZE=str(n))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base.setdefault("OMP_NUM_THREADS", "1")
    procs = [subprocess.Popen(argv, env=dict(base, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n)]
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
                    for q in live:                   # the job is over: end the other ranks (exact pids)
                        q.terminate()
            time.sleep(0.02)
    finally:
        deadline = time.monotonic() + 10.0
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def timed_region(ranks, sync, fn, steps):
    """barrier + device sync, `steps` calls of fn, device sync + barrier; returns MAX elapsed over ranks."""
    sync()
    ranks.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    ranks.barrier()
    return ranks.max_seconds(time.perf_counter() - t0)
