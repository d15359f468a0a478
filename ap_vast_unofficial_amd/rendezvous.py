"""Process-group bootstrap without MPI and without PyTorch: a rank-0 TCP hub over
which the ranks of one node exchange the 128-byte RCCL unique id, meet at host-side barriers and reduce a
few scalars.  The data path never goes through here: filters are gathered by RCCL over xGMI
(`Engine.allgather_filters_dev`).

    rz = Rendezvous.from_env()            # RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, as torchrun sets them
    uid = rz.broadcast(Engine.comm_unique_id() if rz.rank == 0 else None)
    eng.comm_init(uid, rz.rank, rz.world)

The hub listens on MASTER_PORT + APV_RDZV_PORT_OFFSET (default 17): the launcher's own store owns MASTER_PORT.

Wire format.  Nothing is ever unpickled: a value is one tag byte and a fixed or length-prefixed payload --
`N` None, `I` int64, `D` float64, `B` u32 length + raw bytes (at most MAX_BYTES).  A peer opens with a fixed
28-byte hello (magic, 16-byte job token, rank); the hub drops a connection whose hello is malformed, carries
another job's token, names a rank outside 1..world-1 or a rank that has already joined.  The token is the
SHA-256 of APV_RDZV_TOKEN when the launcher sets one, else of "addr:port:world" (which keeps stray
connections and neighbouring jobs apart; it is not a secret).
"""
import hashlib
import os
import socket
import struct
import time

MAGIC = b"APVRDZ2\0"
MAX_BYTES = 4096            # the largest value that crosses this channel is the 128-byte RCCL id


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send(sock, obj):
    if obj is None:
        sock.sendall(b"N")
    elif isinstance(obj, bool):
        raise TypeError("rendezvous carries None, int, float and bytes only")
    elif isinstance(obj, int):
        sock.sendall(b"I" + struct.pack("<q", obj))
    elif isinstance(obj, float):
        sock.sendall(b"D" + struct.pack("<d", obj))
    elif isinstance(obj, (bytes, bytearray)):
        if len(obj) > MAX_BYTES:
            raise ValueError(f"rendezvous: at most {MAX_BYTES} bytes per value")
        sock.sendall(b"B" + struct.pack("<I", len(obj)) + bytes(obj))
    else:
        raise TypeError("rendezvous carries None, int, float and bytes only")


def _recv(sock):
    tag = _recv_exact(sock, 1)
    if tag == b"N":
        return None
    if tag == b"I":
        return struct.unpack("<q", _recv_exact(sock, 8))[0]
    if tag == b"D":
        return struct.unpack("<d", _recv_exact(sock, 8))[0]
    if tag == b"B":
        n = struct.unpack("<I", _recv_exact(sock, 4))[0]
        if n > MAX_BYTES:
            raise ConnectionError("rendezvous: oversized value")
        return _recv_exact(sock, n)
    raise ConnectionError("rendezvous: unknown value tag")


def _job_token(addr, port, world):
    secret = os.environ.get("APV_RDZV_TOKEN")
    text = secret if secret else f"{addr}:{port}:{world}"
    return hashlib.sha256(text.encode()).digest()[:16]


class Rendezvous:
    def __init__(self, rank, world, addr="127.0.0.1", port=29617, timeout=120.0):
        self.rank, self.world = int(rank), int(world)
        self.peers = []          # rank 0: sockets of ranks 1..world-1, by rank
        self.hub = None          # other ranks: socket to rank 0
        if not 0 <= self.rank < max(self.world, 1):
            raise ValueError("rendezvous: rank out of range")
        if self.world == 1:
            return
        token = _job_token(addr, port, self.world)
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            deadline = time.time() + timeout
            by_rank = {}
            try:
                while len(by_rank) < self.world - 1:
                    left = deadline - time.time()
                    if left <= 0:
                        raise TimeoutError(f"rendezvous: {self.world - 1 - len(by_rank)} rank(s) never joined")
                    srv.settimeout(left)
                    conn, _ = srv.accept()
                    try:
                        conn.settimeout(5.0)
                        hello = _recv_exact(conn, len(MAGIC) + 16 + 4)
                        peer = struct.unpack("<I", hello[-4:])[0]
                        ok = (hello[:len(MAGIC)] == MAGIC and hello[len(MAGIC):-4] == token
                              and 0 < peer < self.world and peer not in by_rank)
                    except (OSError, ConnectionError):
                        ok = False
                    if not ok:
                        conn.close()                 # a stray connection, another job, a duplicate: not one of ours
                        continue
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conn.settimeout(timeout)
                    conn.sendall(b"K")               # accepted
                    by_rank[peer] = conn
            except BaseException:
                for c in by_rank.values():
                    c.close()
                raise
            finally:
                srv.close()
            self.peers = [by_rank[r] for r in range(1, self.world)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            s.sendall(MAGIC + token + struct.pack("<I", self.rank))
            if _recv_exact(s, 1) != b"K":
                s.close()
                raise ConnectionError("rendezvous: the hub did not accept this rank")
            self.hub = s

    @classmethod
    def from_env(cls, **kw):
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("MASTER_PORT", "29600")) + int(os.environ.get("APV_RDZV_PORT_OFFSET", "17"))
        return cls(rank, world, addr, port, **kw)

    def gather(self, obj):
        """Rank 0 receives [obj of rank 0, ..., obj of rank world-1]; the others receive None."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            return [obj] + [_recv(p) for p in self.peers]
        _send(self.hub, obj)
        return None

    def broadcast(self, obj):
        """Every rank receives rank 0's obj."""
        if self.world == 1:
            return obj
        if self.rank == 0:
            for p in self.peers:
                _send(p, obj)
            return obj
        return _recv(self.hub)

    def allreduce(self, value, op=max):
        vals = self.gather(value)
        return self.broadcast(op(vals) if self.rank == 0 else None)

    def barrier(self):
        self.allreduce(0)

    def close(self):
        for p in self.peers:
            p.close()
        if self.hub is not None:
            self.hub.close()
        self.peers, self.hub = [], None
